"""Context churn under glibc's heap checks: which step of a renderer's life corrupts the heap when hundreds of contexts come and go
in one process? (tools/soak_camera_grid.py ended twice in `free(): invalid pointer` / a malloc assertion inside the HIP runtime.)
usage (GPU box): MALLOC_CHECK_=3 python -X faulthandler tools/stress_contexts.py MODE COUNT
MODE: create | upload | frames_tree | frames_grids | frames_all | read"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
os.environ.setdefault("UH_HIP_RUNTIME", "system")  # torch-free process: /opt/rocm's HIP runtime (profiles/README.md "The soak crash")
import rust_renderer_amd as rr

mode, count = sys.argv[1], int(sys.argv[2])
scene = rr.scenes.cornell_scene(subdivisions=2, tex_size=16)
rng = np.random.default_rng(1)
for k in range(count):
    W, H = int(rng.integers(40, 200)), int(rng.integers(30, 130))
    r = rr.Renderer(W, H)
    if mode != "create":
        scene.upload(r)
    if mode.startswith("frames") or mode == "read":
        if mode == "frames_tree":
            r.set_option("camera_grid", 0)
            r.set_option("sun_grid", 0)
        loop = rr.FrameLoop(r, scene.make_view(W, H))
        loop.frames(9, rr.PASS_ALL if mode in ("frames_all", "read") else rr.PASS_REFERENCE_PT)
        if mode == "read":
            r.read_accumulation(); r.read_gbuffer_position(); [r.read_reservoirs(i) for i in range(3)]; r.get_stats()
        else:
            r.synchronize()
    del r
    if k % 100 == 99:
        print(mode, k + 1, flush=True)
print(mode, "done", count)
