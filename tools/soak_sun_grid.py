"""Soak for the sun grid (cover depth, margins, fall-back): random sun directions over several scenes, the grid against the tree
walk it replaces - accumulation bit for bit and ray counts - on the GPU. Not part of the test suite.
usage (GPU box): [UH_SOAK_OPTS=name=value,...] python tools/soak_sun_grid.py [first_seed] [count]   (the options go to the grid's renderer)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
os.environ.setdefault("UH_HIP_RUNTIME", "system")  # torch-free process: /opt/rocm's HIP runtime (profiles/README.md "The soak crash")
import rust_renderer_amd as rr

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
scenes = [rr.scenes.sponza_class_scene(detail=0.12, tex_size=32, with_spheres=True, num_lights=0, sphere_subdivisions=2),
          rr.scenes.sponza_class_scene(detail=0.3, tex_size=32, with_spheres=True, num_lights=0, sphere_subdivisions=3),
          rr.scenes.cornell_scene(subdivisions=3, tex_size=16)]
W, H = 160, 90
bad = built = 0
t0 = time.time()
pairs = []
for sc in scenes:
    g, t = sc.upload(rr.Renderer(W, H)), sc.upload(rr.Renderer(W, H))
    t.set_option("sun_grid", 0)
    for kv in filter(None, os.environ.get("UH_SOAK_OPTS", "").split(",")):
        g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    pairs.append((sc, g, t))
for k in range(count):
    rng = np.random.default_rng(first + k)
    sc, g, t = pairs[k % len(pairs)]
    d = rng.normal(size=3)
    if rng.random() < 0.25:
        d[int(rng.integers(0, 3))] = 0.0  # walls edge-on
    if rng.random() < 0.15:
        d = np.sign(d) * (np.abs(d) > np.abs(d).max() - 1e-9)  # axis-aligned
    d = d / np.linalg.norm(d)
    sky = int(rng.integers(0, 2))
    for r in (g, t):
        r.reset_accumulation()
        r.reset_stats()
        loop = rr.FrameLoop(r, sc.make_view(W, H, sun_shadow_enabled=1, sky_enabled=sky, lights_enabled=0))
        loop.view.sun_dir[:] = [float(x) for x in d]
        for _ in range(2):
            loop.frame(rr.PASS_REFERENCE_PT)
    s = g.get_stats()
    ok = bool(np.array_equal(g.read_accumulation().view(np.uint32), t.read_accumulation().view(np.uint32))) and list(s.rays) == list(t.get_stats().rays)
    built += 1 if s.sun_grid_cells else 0
    print("seed %d: %s, sun (%.3f, %.3f, %.3f): cells %d, %d of %d sun rays to the tree: %s (%.0f s)" % (first + k, sc.name, d[0], d[1], d[2], s.sun_grid_cells, s.sun_tree_rays, s.rays[rr.RAY_SUN_SHADOW], "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    bad += 0 if ok else 1
print("sun grid soak: %d directions, %d with a grid, %d mismatches" % (count, built, bad))
sys.exit(1 if bad else 0)
