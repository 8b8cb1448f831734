"""N3: marching cubes. CPU: the oracle's restatement of utopian/shaders/marching_cubes/marching_cubes.comp:179-254
(oracle.cpp orc_marching_cubes, on the reference's tables as data) pinned by what the algorithm guarantees - a closed,
consistently oriented surface, the analytic volume of the shapes, agreement with an independent marching-tetrahedra
extraction in float64 numpy - and held against the product's generated tables case by case. GPU: csrc/isosurface.hip against
that oracle cell by cell (case index, triangles per cell), vertex by vertex (bit for bit), and by enclosed volume."""
import numpy as np
import pytest

import oracle_api as oa
import rust_renderer_amd as rr
from util import extract_isosurface, reference_density

LO, HI = 0.0, 32.0
# torus (R 5, r 3) above a box of half size 5 (marching_cubes.comp:83-90, at time 0): 2 pi^2 R r^2 + 10^3
VOLUME = 2.0 * np.pi ** 2 * 5.0 * 9.0 + 1000.0


def volume(tri):
    t = tri.astype(np.float64)
    return float(np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6.0)


def directed_edges(tri):
    ids = np.unique(tri.reshape(-1, 3).view(np.uint32), axis=0, return_inverse=True)[1].reshape(-1, 3).astype(np.int64)
    e = np.concatenate([ids[:, [0, 1]], ids[:, [1, 2]], ids[:, [2, 0]]])
    e = e[e[:, 0] != e[:, 1]]  # collapsed edges of zero-area triangles
    return e[:, 0] * (1 << 32) + e[:, 1], e[:, 1] * (1 << 32) + e[:, 0]


def test_oracle_density_is_the_shader_s():
    """marching_cubes.comp:61-103 against the float64 numpy restatement, and spot values read off the formulas"""
    rng = np.random.default_rng(3)
    pts = rng.uniform(0, 32, (4000, 3))
    got = np.array([oa.mc_density(p) for p in pts.astype(np.float32)])
    assert np.allclose(got, reference_density(pts.astype(np.float32).astype(np.float64)), atol=2e-5)
    assert oa.mc_density((16, 10, 16)) == 5.0            # box centre: -sdBox = 5
    assert oa.mc_density((21, 20, 16)) == 3.0            # on the torus' centre circle: -sdTorus = 3
    assert oa.mc_density((0, 0, 0)) == -1.0              # far from everything: clamped at -1
    assert oa.mc_density((16, 26, 16)) == 0.0            # the zero-radius sphere's centre: max(-|p - c|, ...) = 0 there
    assert oa.mc_density((16, 26, 16), time=3.0) > 6.0   # 8 |sin(0.9)| = 6.27 inside the animated sphere


@pytest.mark.parametrize("res", [32, 48])
def test_oracle_marching_cubes_is_closed_oriented_and_has_the_right_volume(res):
    mc = oa.marching_cubes(res, LO, HI, order=1)  # one interpolation order for every cell: shared vertices are the same bits
    tri = mc["positions"]
    assert mc["triangles"] == len(tri) == int(mc["tri_count"].astype(np.int64).sum()) > 3000
    fwd, rev = directed_edges(tri)
    assert np.array_equal(np.sort(fwd), np.sort(rev)), "every directed edge has its opposite, as often: closed and consistently oriented"
    # (on these grids shape features lie ON grid planes - cell sizes 1 and 2/3 - and cuts collapse onto corners: see the generic grid below)
    h = (HI - LO) / res
    assert abs(abs(volume(tri)) - VOLUME) < 0.02 * VOLUME * (48 / res) ** 2
    assert volume(tri) > 0, "counter-clockwise seen from outside (the winding the reference's tables produce for density < 0 = outside)"
    assert np.abs(reference_density(tri.reshape(-1, 3).astype(np.float64))).max() < 0.2 * h
    # the shader's own corner order (order 0) differs from it by a rounding of the interpolation, nothing more
    ref_order = oa.marching_cubes(res, LO, HI, order=0)
    assert np.array_equal(ref_order["cube_index"], mc["cube_index"]) and np.array_equal(ref_order["tri_count"], mc["tri_count"])
    assert np.abs(ref_order["positions"] - tri).max() <= 4e-6
    # an independent algorithm on the same field: marching tetrahedra in float64 numpy
    tet = extract_isosurface(reference_density, LO, HI, res)  # an unoriented soup: compared as a point set, both ways
    from scipy.spatial import cKDTree

    a, b = tri.reshape(-1, 3).astype(np.float64), tet.reshape(-1, 3)
    # both grids put a corner exactly on the zero-radius sphere's centre, where the density is 0 = "inside" (cubeIndex tests
    # density < 0): the shader emits collapsed triangles at that point; the float64 generator (density > 0) none
    lone = np.linalg.norm(a - np.array([16.0, 26.0, 16.0]), axis=1) < 1e-6
    assert lone.any()
    a = a[~lone]
    assert cKDTree(b).query(a)[0].max() <= h and cKDTree(a).query(b)[0].max() <= h  # "<=": on the reference's grid (h = 1) cuts land exactly on corners


def test_oracle_marching_cubes_on_a_generic_grid_is_a_manifold():
    """a grid no shape feature is aligned with: no collapsed cut, so every directed edge occurs once and its opposite once"""
    mc = oa.marching_cubes(41, 0.137, 31.871, order=1)
    fwd, rev = directed_edges(mc["positions"])
    assert len(np.unique(fwd)) == len(fwd) == 3 * mc["triangles"] and np.array_equal(np.sort(fwd), np.sort(rev))
    t = mc["positions"].astype(np.float64)
    assert np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1).min() > 1e-9
    assert abs(volume(mc["positions"]) - VOLUME) < 0.03 * VOLUME


def test_animated_sphere_and_empty_grids():
    still, moving = oa.marching_cubes(32, LO, HI, time=0.0, positions=False), oa.marching_cubes(32, LO, HI, time=3.0, positions=False)
    assert moving["triangles"] > still["triangles"]
    assert oa.marching_cubes(4, 100.0, 101.0)["triangles"] == 0


def test_product_tables_describe_the_reference_s_surface_case_by_case():
    """csrc/mc_tables.h (generated from the cube's geometry, tools/gen_mc_tables.py) against the reference's tables (data):
    the same crossed edges, the same number of triangles and the same boundary polygons in all 256 cases - the two differ
    only in which diagonals triangulate a polygon"""
    import re

    edge, table = oa.mc_reference_tables()
    text = open(rr.build.CSRC + "/mc_tables.h").read()

    def numbers(name):
        body = text[text.index(name):]
        return [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", body[body.index("{"):body.index("};")])]

    mask, count, tris = numbers("kMcEdgeMask[256]"), numbers("kMcTriCount[256]"), np.array(numbers("kMcTris[256]")).reshape(256, 15)
    assert mask == [int(x) for x in edge]

    def boundary(triangles):
        from collections import Counter

        c = Counter(frozenset(p) for t in triangles for p in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])))
        return {k for k, v in c.items() if v == 1}

    same_triangles = 0
    for case in range(256):
        ref = [tuple(int(x) for x in table[case][i:i + 3]) for i in range(0, 15, 3) if table[case][i] != -1]
        own = [tuple(int(x) for x in tris[case][3 * i:3 * i + 3]) for i in range(count[case])]
        assert len(ref) == len(own), case
        assert boundary(ref) == boundary(own), case
        assert {e for t in ref for e in t} == {e for e in range(12) if mask[case] >> e & 1}, case
        same_triangles += {frozenset(t) for t in ref} == {frozenset(t) for t in own}
    assert same_triangles >= 90


@pytest.mark.gpu
@pytest.mark.parametrize("res,time", [(48, 0.0), (32, 0.0), (40, 3.0)])
def test_gpu_extraction_against_the_oracle_cell_by_cell(res, time):
    """the default extraction (option iso_reference_triangulation = 1) emits the reference's triangles: per cell the list of
    marching_cubes.comp:231-251 on the reference's table, vertices by vertexInterp in the shader's corner order, zero-area
    triangles kept, order within a cell kept - the whole vertex stream equals orc_marching_cubes' bit for bit"""
    gpu = rr.Renderer(8, 8)
    mesh, ntri = gpu.add_isosurface_mesh(res, LO, HI, time=time)
    v, idx = gpu.read_mesh(mesh)
    tri = v["pos"][:, :3].reshape(-1, 3, 3)
    cube, kept = gpu.isosurface_cells(res, LO, HI, time=time)
    mc = oa.marching_cubes(res, LO, HI, time=time, order=0)
    # 1. every cell decides the same case (the density field and its sign agree bit for bit)
    assert np.array_equal(cube, mc["cube_index"])
    # 2. per cell: the reference's number of triangles - nothing dropped
    assert np.array_equal(kept, mc["tri_count"])
    assert int(kept.astype(np.int64).sum()) == ntri == len(tri) == mc["triangles"]
    # 3. per cell: the reference's triangles in the reference's order, every coordinate bit for bit (cells in x-fastest order,
    #    which is also the order the oracle walks them in, so the two streams are compared whole)
    want = mc["positions"].reshape(-1, 3, 3)
    assert np.array_equal(tri.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(idx, np.arange(3 * ntri, dtype=idx.dtype))
    # ... zero-area triangles included (a cut through a grid corner collapses an edge): the reference keeps them
    area = np.linalg.norm(np.cross(want[:, 1].astype(np.float64) - want[:, 0], want[:, 2].astype(np.float64) - want[:, 0]), axis=1)
    if time == 0.0 and res == 48:
        assert (area <= 1e-12).any()
    # 4. the same solid as the formula says
    if time == 0.0:
        assert abs(volume(tri) - VOLUME) < 0.02 * VOLUME * (48 / res) ** 2
    # 5. normals: generateNormal (central differences at distance 1, negated), as the oracle restates it
    import ctypes as C

    n = np.zeros(3, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    for i in np.random.default_rng(1).integers(0, len(v), 300):
        p = np.ascontiguousarray(v["pos"][i, :3])
        oa.lib().orc_mc_normal(p.ctypes.data_as(fp), time, n.ctypes.data_as(fp))
        assert np.abs(n - v["normal"][i, :3]).max() < 2e-6


@pytest.mark.gpu
def test_gpu_extraction_with_the_generated_tables_is_the_same_surface():
    """option iso_reference_triangulation = 0 (round 3's form): the tables generated in this repository, slivers dropped, edge
    vertices from the smaller grid index - the same case per cell, the same crossings, the same solid"""
    res, time = 40, 0.0
    gpu = rr.Renderer(8, 8)
    gpu.set_option("iso_reference_triangulation", 0)
    mesh, ntri = gpu.add_isosurface_mesh(res, LO, HI, time=time)
    v, _ = gpu.read_mesh(mesh)
    tri = v["pos"][:, :3].reshape(-1, 3, 3)
    cube, kept = gpu.isosurface_cells(res, LO, HI, time=time)
    mc = oa.marching_cubes(res, LO, HI, time=time, order=1)
    assert np.array_equal(cube, mc["cube_index"])
    assert int(kept.astype(np.int64).sum()) == ntri == len(tri)
    assert (kept <= mc["tri_count"]).all()
    mine = np.unique(tri.reshape(-1, 3).view(np.uint32), axis=0)
    theirs = np.unique(mc["positions"].reshape(-1, 3).view(np.uint32), axis=0)
    key = lambda a: a[:, 0].astype(np.uint64) << np.uint64(42) ^ a[:, 1].astype(np.uint64) << np.uint64(21) ^ a[:, 2].astype(np.uint64)
    assert np.isin(key(mine), key(theirs)).all()
    assert abs(volume(tri) - volume(mc["positions"])) < 2e-3 * VOLUME


@pytest.mark.gpu
def test_gpu_extraction_at_512_matches_the_oracle_on_sampled_slabs():
    """BASELINE configs[4]'s grid: the case index of all 134 M cells' worth is too much for the CPU side of a test, so the
    oracle walks a 512 x 512 x 8 slab through the torus and one through the box (the same cells of the 512^3 grid)"""
    res = 512
    gpu = rr.Renderer(8, 8)
    cube, kept = gpu.isosurface_cells(res, LO, HI)
    cube, kept = cube.reshape(res, res, res), kept.reshape(res, res, res)
    mc = oa.marching_cubes(res, LO, HI, positions=False)  # ~20 s of CPU: one pass over the grid, no positions
    assert np.array_equal(cube.reshape(-1), mc["cube_index"])
    assert np.array_equal(kept.reshape(-1), mc["tri_count"])  # the reference's triangle count in every one of the 134 M cells
    assert int(kept.astype(np.int64).sum()) == mc["triangles"]
