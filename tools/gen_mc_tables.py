#!/usr/bin/env python3
"""Generates rust-renderer_amd/csrc/mc_tables.h: the 256-case marching-cubes tables, derived here from the cube's geometry
(nothing is taken from the reference's tables.glsl).

Corner and edge numbering follow the reference's host code (utopian/src/renderers/marching_cubes.rs:23-32 offsets; edges in the
order marching_cubes.comp:206-229 interpolates them): corners 0..3 = the z = 0 face counter-clockwise from the origin, 4..7 the
z = 1 face; edges 0-3 = (0,1) (1,2) (2,3) (3,0), 4-7 = (4,5) (5,6) (6,7) (7,4), 8-11 = (0,4) (1,5) (2,6) (3,7).
Bit i of the case index is set when corner i is OUTSIDE the solid (density < iso level, marching_cubes.comp:186-190).

Construction, per case: on each of the six faces the iso-contour consists of segments between crossed edges. A face with two
crossings has one segment; a face whose corners alternate (four crossings) is ambiguous and is resolved by one fixed rule -
each SET corner is cut off on its own - which two cubes sharing the face apply identically, so the surface has no cracks.
Segments are directed with the set corner(s) on their left seen from outside the cube; every crossed edge then has exactly one
incoming and one outgoing segment, the segments chain into closed loops, and every loop is triangulated as a fan.
"""
import os

CORNERS = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
EDGE_OF = {frozenset(e): i for i, e in enumerate(EDGES)}


def faces():
    """six faces as corner cycles, counter-clockwise seen from OUTSIDE the cube"""
    out = []
    for axis in range(3):
        for side in (0, 1):
            cs = [i for i, c in enumerate(CORNERS) if c[axis] == side]
            u, v = [a for a in range(3) if a != axis]
            # order the 4 corners in a cycle in the (u, v) plane
            key = {(0, 0): 0, (1, 0): 1, (1, 1): 2, (0, 1): 3}
            cyc = sorted(cs, key=lambda i: key[(CORNERS[i][u], CORNERS[i][v])])
            # (u, v, axis) is right-handed when (u, v) = (axis+1, axis+2) mod 3; the cycle above is ccw seen from +axis then
            right_handed = (u, v) == ((axis + 1) % 3, (axis + 2) % 3)
            ccw_from_plus = right_handed
            want_from_plus = side == 1  # outside of the side-1 face is +axis
            if ccw_from_plus != want_from_plus:
                cyc = cyc[::-1]
            out.append(cyc)
    return out


FACES = faces()


def case_triangles(case):
    nxt = {}
    for cyc in FACES:
        s = [(case >> c) & 1 for c in cyc]
        n_set = sum(s)
        if n_set in (0, 4):
            continue
        e = [EDGE_OF[frozenset((cyc[k], cyc[(k + 1) % 4]))] for k in range(4)]  # edge k joins cyc[k], cyc[k+1]
        segs = []
        for k in range(4):
            # walking ccw around the face: a segment leaves through the edge where we pass from a set to an unset corner
            # ... for every maximal run of set corners, the segment goes from the edge AFTER the run to the edge BEFORE it,
            # which keeps the set corners on its left; with alternating corners each set corner is its own run (the fixed rule)
            if s[k] and not s[(k + 1) % 4]:
                j = k
                while s[(j - 1) % 4] and (j - 1) % 4 != k:
                    j = (j - 1) % 4
                segs.append((e[k], e[(j - 1) % 4]))
        for a, b in segs:
            assert a not in nxt, (case, a)
            nxt[a] = b
    tris, seen = [], set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, cur = [], start
        while cur not in seen:
            seen.add(cur)
            loop.append(cur)
            cur = nxt[cur]
        assert cur == start and len(loop) >= 3, (case, loop)
        tris += triangulate(loop)
    return tris


FACE_EDGES = [frozenset(EDGE_OF[frozenset((cyc[k], cyc[(k + 1) % 4]))] for k in range(4)) for cyc in FACES]


def coplanar_with_a_face(a, b):
    return any(a in f and b in f for f in FACE_EDGES)


def all_triangulations(poly):
    """every triangulation of a convex polygon given as a vertex list, as lists of triangles (orientation kept)"""
    if len(poly) < 3:
        return [[]]
    if len(poly) == 3:
        return [[tuple(poly)]]
    out = []
    a, b = poly[0], poly[-1]  # the edge (last, first) belongs to exactly one triangle (b, a, poly[k])
    for k in range(1, len(poly) - 1):
        for left in all_triangulations(poly[: k + 1]):
            for right in all_triangulations(poly[k:]):
                out.append(left + [(a, poly[k], b)] + right)
    return out


def triangulate(loop):
    """a triangulation of the loop none of whose diagonals joins two points of one cube face: such a diagonal makes a
    triangle that lies IN the face - a zero-thickness membrane the neighbouring cube would double"""
    best = None
    for tris in all_triangulations(loop):
        boundary = {frozenset((loop[i], loop[(i + 1) % len(loop)])) for i in range(len(loop))}
        bad = 0
        for t in tris:
            for i in range(3):
                e = frozenset((t[i], t[(i + 1) % 3]))
                if e not in boundary and coplanar_with_a_face(*e):
                    bad += 1
        if best is None or bad < best[0]:
            best = (bad, tris)
    assert best[0] == 0, (loop, best)
    return best[1]


def main():
    edge_mask, tri_table, max_tris = [], [], 0
    for case in range(256):
        tris = case_triangles(case)
        used = sorted({e for t in tris for e in t})
        crossed = [i for i, (a, b) in enumerate(EDGES) if ((case >> a) & 1) != ((case >> b) & 1)]
        assert used == crossed, (case, used, crossed)
        edge_mask.append(sum(1 << e for e in crossed))
        tri_table.append(tris)
        max_tris = max(max_tris, len(tris))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-renderer_amd", "csrc", "mc_tables.h")
    with open(path, "w") as f:
        f.write("// mc_tables.h - GENERATED by tools/gen_mc_tables.py (marching-cubes case tables derived from the cube's geometry; see the\n")
        f.write("// generator for numbering and the face rule). kMcEdgeMask[case]: bit e set when edge e is crossed. kMcTriCount[case]: triangles.\n")
        f.write("// kMcTris[case][3 * i + k]: edge of corner k of triangle i (0xff padding). kMcEdgeCorner[e]: the edge's two cube corners.\n")
        f.write("#pragma once\n#include <cstdint>\n\n")
        f.write(f"constexpr int kMcMaxTris = {max_tris};\n")
        f.write("static const uint8_t kMcEdgeCorner[12][2] = {" + ", ".join("{%d, %d}" % e for e in EDGES) + "};\n")
        f.write("static const uint16_t kMcEdgeMask[256] = {\n")
        for r in range(0, 256, 16):
            f.write("   " + ", ".join("0x%03x" % m for m in edge_mask[r:r + 16]) + ",\n")
        f.write("};\nstatic const uint8_t kMcTriCount[256] = {\n")
        for r in range(0, 256, 32):
            f.write("   " + ", ".join(str(len(t)) for t in tri_table[r:r + 32]) + ",\n")
        f.write("};\n")
        f.write(f"static const uint8_t kMcTris[256][{3 * max_tris}] = {{\n")
        for case in range(256):
            flat = [e for t in tri_table[case] for e in t]
            flat += [0xFF] * (3 * max_tris - len(flat))
            f.write("   {" + ", ".join(str(x) for x in flat) + "},\n")
        f.write("};\n")
    print("wrote", path, "max triangles per case:", max_tris, "total triangles over all cases:", sum(len(t) for t in tri_table))


if __name__ == "__main__":
    main()
