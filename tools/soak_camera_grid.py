"""Soak for the camera grid (perspective margins, distance bounds, fall-back for long lists): random cameras - inside, outside,
touching the geometry, in the plane of surfaces, narrow and wide fields of view, odd frame sizes - over several scenes; the grid
against the tree walk it replaces: accumulation, G-buffer positions and reservoirs bit for bit, ray counts equal. GPU only, not
part of the test suite.
usage (GPU box): python tools/soak_camera_grid.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
os.environ.setdefault("UH_HIP_RUNTIME", "system")  # torch-free process: /opt/rocm's HIP runtime (profiles/README.md "The soak crash")
if os.environ.get("UH_SOAK_BT"):  # (diagnosis: native stack at SIGSEGV / SIGABRT, tools/crash_bt.c)
    import ctypes
    ctypes.CDLL(os.environ["UH_SOAK_BT"]).crash_bt_install()
import rust_renderer_amd as rr
from rust_renderer_amd.camera import Camera

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
scenes = [(rr.scenes.sponza_class_scene(detail=0.12, tex_size=32, with_spheres=True, num_lights=4, sphere_subdivisions=2), (-14, 0, -7), (14, 11, 7)),
          (rr.scenes.sponza_class_scene(detail=0.3, tex_size=32, with_spheres=True, num_lights=0, sphere_subdivisions=3), (-14, 0, -7), (14, 11, 7)),
          (rr.scenes.cornell_scene(subdivisions=3, tex_size=16), (-1, 0, -1), (1, 2, 2.5)),
          (rr.scenes.rtiow_scene(3), (-3, -0.4, -3), (3, 2, 3))]
bad = built = 0
KEEP = []
HELD = []
t0 = time.time()
for k in range(count):
    rng = np.random.default_rng(first + k)
    sc, lo, hi = scenes[k % len(scenes)]
    lo, hi = np.array(lo, dtype=np.float64), np.array(hi, dtype=np.float64)
    W, H = int(rng.integers(40, 200)), int(rng.integers(30, 130))
    eye = lo + rng.random(3) * (hi - lo)
    kind = rng.random()
    if kind < 0.15:
        eye[1] = 0.0                       # in the plane of a floor
    elif kind < 0.3:
        eye = eye + (eye - 0.5 * (lo + hi)) * 4.0  # well outside
    elif kind < 0.4:
        eye = np.round(eye * 2.0) / 2.0     # on round coordinates: in the planes of axis-aligned surfaces
    target = lo + rng.random(3) * (hi - lo)
    if np.linalg.norm(target - eye) < 1e-3:
        target = eye + np.array([1.0, 0.1, 0.2])
    fov = float(rng.choice([20.0, 45.0, 60.0, 90.0, 120.0]))
    cam = Camera(tuple(float(x) for x in eye), tuple(float(x) for x in target), fov, W / H, 0.01, 1000.0)
    sc.camera = cam
    res = []
    for grid_on in (1, 0):
        r = sc.upload(rr.Renderer(W, H))
        r.set_option("camera_grid", grid_on)
        if rng.random() < 0.3 and grid_on and not os.environ.get("UH_SOAK_NO_WALK"):
            r.set_option("camera_grid_max_walk", int(rng.integers(1, 16)))
        for kv in filter(None, os.environ.get("UH_SOAK_OPTS", "").split(",")):
            r.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        loop = rr.FrameLoop(r, sc.make_view(W, H))
        loop.frames(9, rr.PASS_REFERENCE_PT if os.environ.get("UH_SOAK_PASS") == "pt" else rr.PASS_ALL)
        raw = (r.read_accumulation(), r.read_gbuffer_position(), [r.read_reservoirs(i) for i in range(3)])
        res.append((raw[0].copy(), raw[1].copy(), [x.copy() for x in raw[2]], r.get_stats()))
        if os.environ.get("UH_SOAK_HOLD"):
            HELD.append(raw)
        if os.environ.get("UH_SOAK_HOLD"):  # (diagnosis: the arrays the read-backs filled stay allocated for a while)
            HELD.append(res[-1])
            if len(HELD) > 400:
                HELD.pop(0)
        if os.environ.get("UH_SOAK_KEEP"):  # (diagnosis: never destroy a context)
            KEEP.append((r, loop))
        del r
    (a, g, rv, s), (a2, g2, rv2, s2) = res
    ok = np.array_equal(a.view(np.uint32), a2.view(np.uint32)) and np.array_equal(g.view(np.uint32), g2.view(np.uint32)) and all(np.array_equal(x, y) for x, y in zip(rv, rv2)) \
        and list(s.rays) == list(s2.rays)
    built += 1 if s.camera_grid_cells else 0
    print("seed %d: %s %dx%d fov %.0f eye (%.2f, %.2f, %.2f): pixels %d, entries %d, mean list %.1f, %d of %d primary rays to the tree: %s (%.0f s)" % (
        first + k, sc.name, W, H, fov, eye[0], eye[1], eye[2], s.camera_grid_cells, s.camera_grid_entries, s.camera_grid_mean_list, s.camera_tree_rays, s.rays[rr.RAY_PRIMARY],
        "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    bad += 0 if ok else 1
print("camera grid soak: %d cameras, %d with a grid, %d mismatches" % (count, built, bad))
sys.exit(1 if bad else 0)
