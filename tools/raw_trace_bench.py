"""Raw closest-hit traversal kernel on the config-1 scene at 1080p: primary rays once, then the 2.07 M diffuse
bounce rays built from their hits, `reps` times (the profiling target of tools/pmc_passes.sh: the bounce-ray
dispatches are the median of the per-kernel counters).  usage: raw_trace_bench.py [reps] [option=value ...]"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
import rust_renderer_amd as rr

W, H = 1920, 1080
reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
scene = rr.scenes.scene_for_config(1, tex_size=64)
r = rr.Renderer(W, H)
for kv in sys.argv[1:]:
    if '=' in kv:
        k, v = kv.split('=')
        r.set_option(k, int(v))
scene.upload(r)
view = scene.make_view(W, H)
inv_view = np.array(view.inverse_view, dtype=np.float32).reshape(4, 4).T
inv_proj = np.array(view.inverse_projection, dtype=np.float32).reshape(4, 4).T
ys, xs = np.mgrid[0:H, 0:W]
u = (xs.ravel() + 0.5) / W
v = 1.0 - (ys.ravel() + 0.5) / H
d = np.stack([u * 2 - 1, v * 2 - 1, np.ones_like(u), np.ones_like(u)], 1).astype(np.float32)
target = d @ inv_proj.T
t3 = target[:, :3] / np.linalg.norm(target[:, :3], axis=1, keepdims=True)
dirs = t3 @ inv_view[:3, :3].T
org = np.broadcast_to(inv_view[:3, 3], dirs.shape)
rays = np.empty((W * H, 8), np.float32)
rays[:, 0:3] = org; rays[:, 3] = 0.001; rays[:, 4:7] = dirs; rays[:, 7] = 10000.0
r.set_option("time_kernels", 1)


def timed(rays, label, reps):
    best = 1e9
    for _ in range(reps):
        r.reset_stats()
        out = r.trace_closest(rays)
        best = min(best, r.get_stats().trace_closest_ms)
    print("%-24s %8.3f ms  %7.1f Mrays/s" % (label, best, len(rays) / best / 1e3), flush=True)
    return out


tuv, mesh, prim = timed(rays, "primary, pixel order", 1)
hit = tuv[:, 0] > 0
P = org[hit] + tuv[hit, 0:1] * dirs[hit]
rng = np.random.default_rng(1)
rnd = rng.normal(size=P.shape).astype(np.float32)
rnd /= np.linalg.norm(rnd, axis=1, keepdims=True)
rnd = np.where((np.sum(rnd * -dirs[hit], 1) < 0)[:, None], -rnd, rnd)
b = np.empty((len(P), 8), np.float32)
b[:, 0:3] = P + 1e-3 * rnd; b[:, 3] = 0.001; b[:, 4:7] = rnd; b[:, 7] = 10000.0
out_b = timed(b, "bounce, pixel order", reps)
np.save("gpurun_out/raw_bounce_hits.npy", out_b[0])
