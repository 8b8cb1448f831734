"""How much does the ORDER of incoherent bounce rays in the queue matter to k_trace_closest_raw?
Builds diffuse bounce rays from the primary hits of the config-2 scene on the host, then times the raw
closest-hit kernel over the same ray set in different orders."""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
import rust_renderer_amd as rr

W, H = 1920, 1080
scene = rr.scenes.scene_for_config(1, tex_size=64)
r = rr.Renderer(W, H)
for kv in sys.argv[1:]:
    k, v = kv.split('=')
    r.set_option(k, int(v))
scene.upload(r)
cam = scene.camera
cam.aspect_ratio = W / H
view = scene.make_view(W, H)
inv_view = np.array(view.inverse_view, dtype=np.float32).reshape(4, 4).T  # column-major storage
inv_proj = np.array(view.inverse_projection, dtype=np.float32).reshape(4, 4).T
ys, xs = np.mgrid[0:H, 0:W]
u = (xs.ravel() + 0.5) / W
v = 1.0 - (ys.ravel() + 0.5) / H
d = np.stack([u * 2 - 1, v * 2 - 1, np.ones_like(u), np.ones_like(u)], 1).astype(np.float32)
target = d @ inv_proj.T
t3 = target[:, :3] / np.linalg.norm(target[:, :3], axis=1, keepdims=True)
dirs = t3 @ inv_view[:3, :3].T
org = np.broadcast_to(inv_view[:3, 3], dirs.shape)
n = W * H
rays = np.empty((n, 8), np.float32)
rays[:, 0:3] = org; rays[:, 3] = 0.001; rays[:, 4:7] = dirs; rays[:, 7] = 10000.0


def timed(rays, label, reps=3):
    r.set_option("time_kernels", 1)
    best = 1e9
    for _ in range(reps):
        r.reset_stats()
        out = r.trace_closest(rays)
        best = min(best, r.get_stats().trace_closest_ms)
    print("%-34s %8.3f ms  %7.1f Mrays/s" % (label, best, len(rays) / best / 1e3), flush=True)
    return out


tuv, mesh, prim = timed(rays, "primary, pixel order")
hit = tuv[:, 0] > 0
P = org[hit] + tuv[hit, 0:1] * dirs[hit]
rng = np.random.default_rng(1)
# diffuse bounce: direction = -incoming flipped about a random unit vector in the hemisphere facing the camera
rnd = rng.normal(size=P.shape).astype(np.float32)
rnd /= np.linalg.norm(rnd, axis=1, keepdims=True)
facing = -dirs[hit]
rnd = np.where((np.sum(rnd * facing, 1) < 0)[:, None], -rnd, rnd)   # crude: hemisphere about the view vector
b = np.empty((len(P), 8), np.float32)
b[:, 0:3] = P + 1e-3 * rnd; b[:, 3] = 0.001; b[:, 4:7] = rnd; b[:, 7] = 10000.0
print("bounce rays:", len(b))


def morton(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = ((p - lo) / (hi - lo + 1e-9) * ((1 << bits) - 1)).astype(np.uint64)
    code = np.zeros(len(p), np.uint64)
    for i in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(i)) & np.uint64(1)) << np.uint64(3 * i + a)
    return code


octant = ((b[:, 4] < 0).astype(np.uint64) | ((b[:, 5] < 0).astype(np.uint64) << np.uint64(1)) | ((b[:, 6] < 0).astype(np.uint64) << np.uint64(2)))
mc = morton(b[:, 0:3])
timed(b, "bounce, pixel order")
timed(b[rng.permutation(len(b))], "bounce, random shuffle")
timed(b[np.argsort(octant, kind='stable')], "bounce, octant bins (stable)")
timed(b[np.argsort(mc, kind='stable')], "bounce, morton(origin)")


def per_xcd(key8, label, chunk=64, blocks=256 * 6, waves_per_block=4):
    """chunks of 64 rays whose 3-bit key equals the XCD their wave runs on: raw rays are handed out statically, chunk c to wave
    c % num_waves, block = wave // 4, XCD = block % 8 (round-robin dispatch)"""
    num_waves = blocks * waves_per_block
    pools = [list(np.flatnonzero(key8 == k)) for k in range(8)]
    pos = [0] * 8
    order = np.empty(len(b), np.int64)
    nchunks = (len(b) + chunk - 1) // chunk
    at = 0
    for c in range(nchunks):
        want = ((c % num_waves) // waves_per_block) % 8
        need = min(chunk, len(b) - at)
        while need:
            k = want if pos[want] < len(pools[want]) else max(range(8), key=lambda q: len(pools[q]) - pos[q])
            take = min(need, len(pools[k]) - pos[k])
            order[at:at + take] = pools[k][pos[k]:pos[k] + take]
            pos[k] += take
            at += take
            need -= take
    timed(b[order], label)


per_xcd(octant.astype(np.int64), "bounce, octant = XCD, 64-ray chunks")
per_xcd((rng.integers(0, 8, len(b))).astype(np.int64), "bounce, random key = XCD (control)")
zq = np.minimum(((b[:, 0] - b[:, 0].min()) / (np.ptp(b[:, 0]) + 1e-9) * 8).astype(np.int64), 7)
per_xcd(zq, "bounce, origin x slab = XCD")
timed(b[np.argsort((octant << np.uint64(30)) | mc, kind='stable')], "bounce, octant then morton")
timed(b[np.argsort((mc >> np.uint64(15) << np.uint64(3)) | octant, kind='stable')], "bounce, morton15 then octant")
# direction-only fine sort: octahedral-ish quantisation of direction, 6 bits per axis
dq = morton(b[:, 4:7], bits=5)
timed(b[np.argsort((mc >> np.uint64(18) << np.uint64(15)) | dq, kind='stable')], "bounce, morton12 then dir15")
timed(b[np.argsort((dq << np.uint64(30)) | mc, kind='stable')], "bounce, dir15 then morton")

# ---- per-ray visit counts -> wave utilisation model ------------------------------------------------
r.set_option("raw_visit_counts", 1)
for label, rs in (("primary", rays), ("bounce", b)):
    tuv2, _, _ = r.trace_closest(rs)
    nodes = tuv2[:, 1].astype(np.int64); tris = tuv2[:, 2].astype(np.int64)
    L = nodes + tris
    np.save("gpurun_out/visits_%s.npy" % label, np.stack([nodes, tris], 1).astype(np.int16))
    n64 = len(L) // 64 * 64
    Lw = L[:n64].reshape(-1, 64)
    iters = Lw.max(1)
    print("%s: nodes/ray %.2f tris/ray %.2f  mean L %.1f  p50 %d p90 %d p99 %d max %d | wave max mean %.1f  slot utilisation %.3f" % (
        label, nodes.mean(), tris.mean(), L.mean(), np.percentile(L, 50), np.percentile(L, 90), np.percentile(L, 99), L.max(), iters.mean(), L[:n64].sum() / (64.0 * iters.sum())))
    for K in (16, 24, 32, 40, 48):
        # pass p handles rays with L > p*K, compacted in order into waves of 64, each running min(K, remaining max)
        total_iters = 0
        rem = L[:n64].copy()
        passes = 0
        while len(rem):
            m = len(rem) // 64 * 64
            w = rem[:m].reshape(-1, 64) if m else np.zeros((0, 64), np.int64)
            total_iters += np.minimum(w.max(1), K).sum() if m else 0
            if len(rem) > m:
                total_iters += min(rem[m:].max(), K)
            rem = rem[rem > K] - K
            passes += 1
        print("   cap K=%d: passes %d, wave-iterations %.3f of uncapped" % (K, passes, total_iters / iters.sum()))
