"""Soak: many random triangle soups (every material type, mirrored and sheared instances, lights), HIP path against the CPU
oracle - closest-hit and any-hit queries bit for bit over a few hundred thousand rays per scene, then two full frames
(every pass) bit for bit with the sky off. Not part of the test suite: a one-off confidence run after kernel changes.
usage (GPU box): python tools/soak_parity.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
os.environ.setdefault("UH_HIP_RUNTIME", "system")  # torch-free process: /opt/rocm's HIP runtime (profiles/README.md "The soak crash")
import oracle_api as oa
import rust_renderer_amd as rr
from util import random_rays, run_frames
from test_gpu_parity import _soup_scene

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 12)
W, H = 128, 72
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    scene = _soup_scene(seed)
    gpu = scene.upload(rr.Renderer(W, H))
    gpu.set_option("fused_bounces", -1 if seed % 2 else 0)  # odd seeds: every frame's later bounces in the fused kernel; even: the wavefront of launches
    cpu = scene.upload(oa.OracleRenderer(W, H))
    rays = random_rays(((-3, -1, -3), (3, 3, 3)), 150000, seed=seed)
    ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(gpu.trace_closest(rays), cpu.trace_closest(rays)))
    ok &= bool(np.array_equal(gpu.trace_any(rays), cpu.trace_any(rays)))
    for r in (gpu, cpu):
        run_frames(r, scene, W, H, 2, rr.PASS_ALL if scene.lights else rr.PASS_REFERENCE_PT, sky_enabled=0)
    ok &= bool(np.array_equal(gpu.read_accumulation().view(np.uint32), cpu.read_accumulation().view(np.uint32)))
    ok &= list(gpu.get_stats().rays)[:4] == list(cpu.get_stats().rays)[:4]
    for which in range(3):
        ok &= bool(np.array_equal(gpu.read_reservoirs(which), cpu.read_reservoirs(which)))
    # and through the batched path
    gpu2 = scene.upload(rr.Renderer(W, H))
    rr.FrameLoop(gpu2, scene.make_view(W, H, sky_enabled=0)).frames(2, rr.PASS_ALL if scene.lights else rr.PASS_REFERENCE_PT)
    ok &= bool(np.array_equal(gpu2.read_accumulation().view(np.uint32), cpu.read_accumulation().view(np.uint32)))
    print("seed %d: %d triangles, %s (%.0f s)" % (seed, scene.num_triangles, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    bad += 0 if ok else 1
print("soak: %d scenes, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
