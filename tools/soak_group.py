"""Soak for the in-process group (uh_mgpu_*: tiles for the path tracer, bands of rows for the reservoir passes, peer-copy
exchange): random triangle soups with lights, 2-5 contexts on GPU 0, odd frame sizes, batched frames - accumulation, all
three reservoir buffers and ray counts against one context, bit for bit. Not part of the test suite.
usage (GPU box): python tools/soak_group.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
os.environ.setdefault("UH_HIP_RUNTIME", "system")  # torch-free process: /opt/rocm's HIP runtime (profiles/README.md "The soak crash")
import rust_renderer_amd as rr
from test_gpu_parity import _soup_scene

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 12)
rng = np.random.default_rng(first)
bad = done = 0
t0 = time.time()
seed = first
while done < count:
    scene = _soup_scene(seed)
    seed += 1
    if not scene.lights:
        continue
    done += 1
    W, H = int(rng.integers(40, 160)), int(rng.integers(20, 130))
    n, tile, frames = int(rng.integers(2, 6)), int(rng.choice([8, 16, 32, 64])), int(rng.integers(2, 22))
    one = scene.upload(rr.Renderer(W, H))
    group = scene.upload(rr.MultiGpuRenderer(W, H, devices=[0] * n, tile_size=tile))
    for r in (one, group):
        rr.FrameLoop(r, scene.make_view(W, H, use_ris_light_sampling=1)).frames(frames, rr.PASS_ALL)
    ga, oa = group.read_accumulation().view(np.uint32), one.read_accumulation().view(np.uint32)
    ok = bool(np.array_equal(ga, oa))
    detail = [] if ok else ["accumulation: %d of %d pixels, rows %s" % (int((ga != oa).any(axis=-1).sum()), W * H, sorted(set(np.nonzero((ga != oa).any(axis=-1))[0].tolist()))[:12])]
    for which in range(3):
        gr, orr = group.read_reservoirs(which), one.read_reservoirs(which)
        same = gr.tobytes() == orr.tobytes()
        if not same:
            gb, ob = np.frombuffer(gr.tobytes(), np.uint8).reshape(H * W, -1), np.frombuffer(orr.tobytes(), np.uint8).reshape(H * W, -1)
            rows = sorted(set((np.nonzero((gb != ob).any(axis=1))[0] // W).tolist()))
            detail.append("reservoirs[%d]: %d pixels, rows %s" % (which, int((gb != ob).any(axis=1).sum()), rows[:12]))
        ok &= same
    if list(group.get_stats().rays) != list(one.get_stats().rays):
        ok = False
        detail.append("rays %s against %s" % (list(group.get_stats().rays), list(one.get_stats().rays)))
    print("seed %d: %dx%d, %d contexts, tile %d, %d frames, %d triangles, %d lights: %s (%.0f s)" % (seed - 1, W, H, n, tile, frames, scene.num_triangles, len(scene.lights), "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    bad += 0 if ok else 1
    for d in detail:
        print("      " + d, flush=True)
    one.close()
    group.close()
print("group soak: %d scenes, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
