"""Does it pay to give each XCD the rays of ONE region of the scene? (VERDICT r4 do-this 3b.) The closest-hit kernel's records cost
4.4 clk per CU from an XCD's 4 MiB L2 and 8-10 beyond it (profiles/r02_microbench_rates.txt), its L2 hit rate is 0.75, and the tree +
packets (22 MB) would fit the eight L2s together if every XCD only ever saw an eighth of the scene. No kernel change is needed to
find out: uh_trace_closest hands chunk c (64 rays) to wave c % num_waves, i.e. to block (c % num_waves) // 4, and blocks go round the
XCDs (block b -> XCD b % 8) - so the ORDER of the rays decides which XCD traces which ray. Bounce rays of the config-1 scene at 1080p
(2.07 M, diffuse directions from the primary hits), four orders:
   pixel     as the frame produces them (a wave's rays leave 64 neighbouring pixels; an XCD sees the whole screen)
   random    a random permutation (deep bounces: no coherence inside a wave either)
   regions   the origins split into 8 regions of equal count (median splits, 3 levels); XCD x gets the rays of region x only,
             rays inside a region in random order
   regions+pixel   the same, rays inside a region in pixel order
usage: xcd_locality_experiment.py [reps]"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
import rust_renderer_amd as rr

W, H = 1920, 1080
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scene = rr.scenes.scene_for_config(1, tex_size=64)
r = rr.Renderer(W, H)
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


def timed(rays, label, reps, want=None):
    best = 1e9
    for _ in range(reps):
        r.reset_stats()
        out = r.trace_closest(rays)
        best = min(best, r.get_stats().trace_closest_ms)
    print("%-28s %8.3f ms  %7.1f Mrays/s" % (label, best, len(rays) / best / 1e3), flush=True)
    return out, best


(tuv, mesh, prim), _ = timed(rays, "primary, pixel order", 1)
hit = tuv[:, 0] > 0
P = org[hit] + tuv[hit, 0:1] * dirs[hit]
rng = np.random.default_rng(1)
rnd = rng.normal(size=P.shape).astype(np.float32)
rnd /= np.linalg.norm(rnd, axis=1, keepdims=True)
rnd = np.where((np.sum(rnd * -dirs[hit], 1) < 0)[:, None], -rnd, rnd)
b = np.empty((len(P), 8), np.float32)
b[:, 0:3] = P + 1e-3 * rnd; b[:, 3] = 0.001; b[:, 4:7] = rnd; b[:, 7] = 10000.0
n = len(b) // 512 * 512  # whole chunks for every XCD
b = b[:n]

# the kernel's raw-ray grid: min(needed, num_cus * 5 blocks) blocks of 4 waves (kernels.hip launch_trace_closest_raw)
blocks = min((n + 255) // 256, 256 * 5)
num_waves = blocks * 4
chunk = np.arange(n // 64)
xcd_of_chunk = ((chunk % num_waves) // 4) % 8


def kd_regions(pts, levels=3):
    idx = [np.arange(len(pts))]
    for _ in range(levels):
        nxt = []
        for ix in idx:
            p = pts[ix]
            ax = int(np.argmax(p.max(0) - p.min(0)))
            order = ix[np.argsort(p[:, ax], kind="stable")]
            nxt += [order[: len(order) // 2], order[len(order) // 2:]]
        idx = nxt
    return idx


def arrange(region_lists):
    """rays of region x into the chunks XCD x traces (region lists may differ in length: what is left over fills the remaining chunks)"""
    out = np.empty(n, dtype=np.int64)
    cursors = [0] * 8
    left = []
    for c in chunk:
        x = xcd_of_chunk[c]
        lst = region_lists[x]
        if cursors[x] + 64 <= len(lst):
            out[c * 64:(c + 1) * 64] = lst[cursors[x]:cursors[x] + 64]
            cursors[x] += 64
        else:
            left.append(c)
    rest = np.concatenate([region_lists[x][cursors[x]:] for x in range(8)])
    for k, c in enumerate(left):
        out[c * 64:(c + 1) * 64] = rest[k * 64:(k + 1) * 64]
    assert len(np.unique(out)) == n
    return out


ref, _ = timed(b, "bounce, pixel order (warm-up)", 2)
orders = {"pixel": np.arange(n), "random": rng.permutation(n)}
regions = kd_regions(b[:, 0:3])
orders["regions"] = arrange([rng.permutation(ix) for ix in regions])
orders["regions+pixel"] = arrange([np.sort(ix) for ix in regions])
# control: the same chunk structure with the rays dealt to the XCDs at random (each XCD sees every region)
mixed = np.concatenate(regions)[rng.permutation(n)]
orders["control"] = arrange([mixed[k * (n // 8):(k + 1) * (n // 8)] for k in range(8)])
# 64 regions, 8 per XCD: finer locality inside an XCD's stream
regions64 = kd_regions(b[:, 0:3], levels=6)
orders["regions64"] = arrange([np.concatenate([rng.permutation(ix) for ix in regions64[8 * x:8 * x + 8]]) for x in range(8)])
# 64 regions dealt round-robin to the XCDs chunk by chunk (locality inside a wave, none per XCD)
orders["waves_local"] = np.concatenate([rng.permutation(ix) for ix in regions64])
orders["random_b"] = rng.permutation(n)  # (a second permutation, measured at another place in the round)
mixed = rng.permutation(n)
orders["control_b"] = arrange([mixed[k * (n // 8):(k + 1) * (n // 8)] for k in range(8)])
orders["random_c"] = orders["random"][::-1].copy()
times = {k: [] for k in orders}
for rep in range(reps):  # round-robin over the orders: clock drift and neighbours hit every order alike
    for k, order in orders.items():
        r.reset_stats()
        out = r.trace_closest(b[order])
        times[k].append(r.get_stats().trace_closest_ms)
        if rep == 0:
            assert np.array_equal(out[0].view(np.uint32), ref[0][order].view(np.uint32)), "hits do not depend on the order"
for k, t in times.items():
    print("%-16s min %.3f  median %.3f ms   (%s)" % (k, min(t), float(np.median(t)), " ".join("%.3f" % x for x in t)))
