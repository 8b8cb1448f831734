"""Writes the input of tools/bvh_visits.cpp: the config-1 scene's world-space triangles and a set of rays (primary rays of a
small frame traced by the CPU oracle, then diffuse bounce rays from their hits), and runs the study."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_api as oa
import rust_renderer_amd as rr

W, H = 320, 180
scene = rr.scenes.scene_for_config(1, tex_size=16)
o = scene.upload(oa.OracleRenderer(W, H))
corners = []
for model, transform in scene.models:
    for m in model.meshes:
        t = (m.transform if transform is None else rr.api.compose3x4(transform, m.transform)).reshape(3, 4).astype(np.float32)
        p = m.vertices["pos"][:, :3][m.indices.reshape(-1, 3)]  # (nt, 3, 3)
        corners.append((p @ t[:, :3].T + t[:, 3]).astype(np.float32).reshape(-1, 9))
corners = np.concatenate(corners)
view = scene.make_view(W, H)
inv_view = np.array(view.inverse_view, dtype=np.float32).reshape(4, 4).T
inv_proj = np.array(view.inverse_projection, dtype=np.float32).reshape(4, 4).T
ys, xs = np.mgrid[0:H, 0:W]
u = (xs.ravel() + 0.5) / W; v = 1.0 - (ys.ravel() + 0.5) / H
d = np.stack([u * 2 - 1, v * 2 - 1, np.ones_like(u), np.ones_like(u)], 1).astype(np.float32)
target = d @ inv_proj.T
t3 = target[:, :3] / np.linalg.norm(target[:, :3], axis=1, keepdims=True)
dirs = t3 @ inv_view[:3, :3].T
org = np.broadcast_to(inv_view[:3, 3], dirs.shape)
rays = np.empty((W * H, 8), np.float32)
rays[:, 0:3] = org; rays[:, 3] = 0.001; rays[:, 4:7] = dirs; rays[:, 7] = 10000.0
tuv, mesh, prim = o.trace_closest(rays)
hit = tuv[:, 0] > 0
P = org[hit] + tuv[hit, 0:1] * dirs[hit]
rng = np.random.default_rng(1)
rnd = rng.normal(size=P.shape).astype(np.float32); rnd /= np.linalg.norm(rnd, axis=1, keepdims=True)
rnd = np.where((np.sum(rnd * -dirs[hit], 1) < 0)[:, None], -rnd, rnd)
b = np.empty((len(P), 8), np.float32)
b[:, 0:3] = P + 1e-3 * rnd; b[:, 3] = 0.001; b[:, 4:7] = rnd; b[:, 7] = 10000.0
exe = "/tmp/bvh_visits"
flags = [a for a in sys.argv[1:] if a.startswith("-D")]  # e.g. -DUH_BVH_CTRI=0.5f: the builder's cost of a triangle slot against a node visit
subprocess.run(["g++", "-O2", "-std=c++17"] + flags + ["-I", os.path.join(ROOT, "rust-renderer_amd/csrc"), os.path.join(ROOT, "tools/bvh_visits.cpp"),
                os.path.join(ROOT, "rust-renderer_amd/csrc/bvh_build.cpp"), "-o", exe, "-pthread"], check=True)
for name, rs in (("primary", rays), ("bounce", b)):
    path = f"/tmp/bvh_visits_{name}.bin"
    with open(path, "wb") as f:
        f.write(np.uint32(len(corners)).tobytes()); f.write(np.uint32(len(rs)).tobytes()); f.write(corners.tobytes()); f.write(rs.tobytes())
    print(f"== {name}: {len(rs)} rays, {len(corners)} triangles", flush=True)
    subprocess.run([exe, path], check=True)
