"""CPU: the generated marching-cubes tables (tools/gen_mc_tables.py -> rust-renderer_amd/csrc/mc_tables.h; SURVEY.md section 8f N3,
reference shaders/marching_cubes/marching_cubes.comp:179-254 uses its own tables.glsl, which is not taken):
structural properties per case and watertightness of the surface they produce on a grid."""
import collections
import itertools
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EC = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
C = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]


def load_tables():
    src = open(os.path.join(ROOT, "rust-renderer_amd", "csrc", "mc_tables.h")).read()

    def nums(name):
        return [int(x, 0) for x in re.findall(r"0x[0-9a-f]+|\d+", src[src.index(name):].split("{", 1)[1].split("};")[0])]

    maxt = int(re.search(r"kMcMaxTris = (\d+)", src).group(1))
    return np.array(nums("kMcEdgeMask[256]")), np.array(nums("kMcTriCount[256]")), np.array(nums("kMcTris[256]")).reshape(256, 3 * maxt), maxt


def test_generator_reproduces_the_committed_header(tmp_path):
    before = open(os.path.join(ROOT, "rust-renderer_amd", "csrc", "mc_tables.h")).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_mc_tables.py")], check=True, capture_output=True)
    assert open(os.path.join(ROOT, "rust-renderer_amd", "csrc", "mc_tables.h")).read() == before


def test_cases_use_exactly_the_crossed_edges():
    mask, cnt, tris, maxt = load_tables()
    assert maxt == 5 and cnt[0] == 0 and cnt[255] == 0
    for case in range(256):
        crossed = {e for e, (a, b) in enumerate(EC) if ((case >> a) & 1) != ((case >> b) & 1)}
        assert mask[case] == sum(1 << e for e in crossed)
        used = set(tris[case, : 3 * cnt[case]].tolist())
        assert used == crossed and (tris[case, 3 * cnt[case]:] == 0xFF).all()


def surface_edges(field):
    _, cnt, tris, _ = load_tables()
    n = field.shape[0] - 1
    use = collections.Counter()
    for x, y, z in itertools.product(range(n), repeat=3):
        case = sum(1 << i for i, (a, b, c) in enumerate(C) if field[x + a, y + b, z + c] < 0)
        for t in range(cnt[case]):
            vs = []
            for k in range(3):
                a, b = EC[tris[case, 3 * t + k]]
                vs.append(tuple(sorted(((x + C[a][0], y + C[a][1], z + C[a][2]), (x + C[b][0], y + C[b][1], z + C[b][2])))))
            for k in range(3):
                use[(vs[k], vs[(k + 1) % 3])] += 1
    return use, n


def test_surface_is_closed_and_consistently_oriented():
    """every mesh edge away from the grid boundary is used by exactly two triangles, in opposite directions - also across
    ambiguous faces (a random sign field is full of them) - and no triangle lies inside a cell face"""
    rng = np.random.default_rng(0)
    g = np.indices((11, 11, 11)).astype(float)
    for field in (rng.normal(size=(8, 8, 8)), np.sqrt(((g - 4.8) ** 2).sum(0)) - 3.7):
        use, n = surface_edges(field)
        assert use

        def on_boundary(v):
            p, q = v
            return any(p[i] == q[i] and p[i] in (0, n) for i in range(3))

        for (a, b), c in use.items():
            if on_boundary(a) and on_boundary(b):
                continue
            assert c == 1 and use.get((b, a), 0) == 1, (a, b, c)
