"""CPU: host-side logic above the C ABI - camera matrices (glam semantics), scene generators,
the frame protocol, tile index math."""
import hashlib

import numpy as np

import rust_renderer_amd as rr
from rust_renderer_amd import camera as cam


def test_look_at_and_perspective_match_glam_formulas():
    eye, target = np.float32([-10.28, 2.10, -0.18]), np.float32([0.0, 0.5, 0.0])
    v = cam.look_at_rh(eye, target, (0, 1, 0))
    assert np.allclose(v @ np.append(eye, 1.0), [0, 0, 0, 1], atol=1e-5), "the eye maps to the view-space origin"
    f = (target - eye) / np.linalg.norm(target - eye)
    assert np.allclose((v @ np.append(eye + f, 1.0))[:3], [0, 0, -1], atol=1e-5), "right-handed: forward is -z"
    assert np.allclose(v[:3, :3] @ v[:3, :3].T, np.eye(3), atol=1e-6)
    p = cam.perspective_rh(np.radians(60.0), 16 / 9, 0.01, 1000.0)
    h = 1.0 / np.tan(np.radians(30.0))
    assert np.allclose([p[0, 0], p[1, 1], p[3, 2]], [h / (16 / 9), h, -1.0], rtol=1e-6)
    near = p @ np.float32([0, 0, -0.01, 1])
    far = p @ np.float32([0, 0, -1000.0, 1])
    assert abs(near[2] / near[3]) < 1e-6 and abs(far[2] / far[3] - 1.0) < 1e-5, "depth range 0..1"
    assert np.allclose(cam.inverse(p) @ p, np.eye(4), atol=1e-4)
    assert np.array_equal(cam.to_glam(p).reshape(4, 4).T, p), "column-major storage"


def test_default_view_matches_reference_defaults():
    scene = rr.scenes.cornell_scene(1, 4)
    v = rr.default_view(scene.camera, 200, 100, num_lights=3)
    # prototype/src/main.rs:55-86
    assert (v.samples_per_frame, v.total_samples, v.num_bounces) == (1, 0, 5)
    assert (v.sky_enabled, v.sun_shadow_enabled, v.lights_enabled, v.max_num_lights_used) == (1, 1, 1, 10000)
    assert (v.temporal_reuse_enabled, v.spatial_reuse_enabled, v.accumulation_limit, v.use_ris_light_sampling) == (1, 1, 999999, 1)
    assert np.allclose(v.sun_dir[:], np.float32([0.0, 0.9, 0.15]) / np.linalg.norm([0.0, 0.9, 0.15]))
    assert list(v.prev_frame_projection_view[:]) == [-1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, -1]


class _Recorder:
    backend = "hip"

    def __init__(self):
        self.calls = []

    def get_num_lights(self):
        return 7

    def render_frame(self, view, mask):
        self.calls.append((view.total_samples, view.num_lights, list(view.prev_frame_projection_view[:]), mask))

    def reset_accumulation(self):
        self.calls.append("reset")


def test_frame_loop_protocol():
    scene = rr.scenes.cornell_scene(1, 4)
    rec = _Recorder()
    loop = rr.FrameLoop(rec, scene.make_view(32, 32, samples_per_frame=2))
    loop.frame()
    loop.frame(rr.PASS_REFERENCE_PT)
    (ts1, nl1, pv1, m1), (ts2, nl2, pv2, m2) = rec.calls
    assert (ts1, ts2) == (2, 4), "total_samples += samples_per_frame BEFORE the frame (main.rs:467-469)"
    assert nl1 == 7 and m1 == rr.PASS_ALL and m2 == rr.PASS_REFERENCE_PT
    assert pv1[0] == -1.0, "first frame sees the initial -identity"
    proj = np.array(loop.view.projection[:], dtype=np.float32).reshape(4, 4).T
    view = np.array(loop.view.view[:], dtype=np.float32).reshape(4, 4).T
    assert np.allclose(np.float32(pv2).reshape(4, 4).T, proj @ view, atol=1e-5), "prev = projection * view AFTER the frame (main.rs:545-546)"
    loop.reset()
    assert loop.view.total_samples == 0 and rec.calls[-1] == "reset"


def _digest(scene):
    h = hashlib.sha256()
    for model, _ in scene.models:
        for m in model.meshes:
            h.update(m.vertices.tobytes())
            h.update(m.indices.tobytes())
        for t in model.textures:
            h.update(t.tobytes())
    h.update(np.float32(scene.lights).tobytes())
    return h.hexdigest()


def test_scene_generators_are_deterministic_and_sized():
    a = rr.scenes.sponza_class_scene(detail=0.1, tex_size=8, num_lights=16)
    b = rr.scenes.sponza_class_scene(detail=0.1, tex_size=8, num_lights=16)
    assert _digest(a) == _digest(b)
    assert a.num_meshes == 103 and len(a.models[0][0].textures) == 25 and len(a.lights) == 16
    full = rr.scenes.sponza_class_scene(detail=1.0, tex_size=4)
    assert full.num_meshes == 103 and abs(full.num_triangles - 262267) / 262267 < 0.005, full.num_triangles
    r = rr.scenes.rtiow_scene(2)
    assert r.num_triangles == 4 * 20 * 16 and [m.material_type for m in r.models[0][0].meshes] == [0, 0, 2, 1]


def test_scene_geometry_is_well_formed():
    s = rr.scenes.sponza_class_scene(detail=0.1, tex_size=8, with_spheres=True)
    for model, _ in s.models:
        for m in model.meshes:
            assert m.indices.max() < len(m.vertices) and len(m.indices) % 3 == 0
            n = m.vertices["normal"][:, :3]
            assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-3)
            assert np.isfinite(m.vertices["pos"]).all() and np.isfinite(m.vertices["uv"]).all()


def test_icosphere_is_closed_and_unit():
    v, i = rr.scenes.icosphere(2)
    assert len(i) // 3 == 20 * 16
    assert np.allclose(np.linalg.norm(v["pos"][:, :3], axis=1), 1.0, atol=1e-6)
    edges = np.sort(np.concatenate([i.reshape(-1, 3)[:, [0, 1]], i.reshape(-1, 3)[:, [1, 2]], i.reshape(-1, 3)[:, [2, 0]]]), axis=1)
    _, counts = np.unique(edges, axis=0, return_counts=True)
    assert (counts == 2).all(), "every edge is shared by exactly two triangles"


def test_transform_helpers():
    t = rr.transform3x4((2, 3, 4), (5, 6, 7))
    assert list(t) == [2, 0, 0, 5, 0, 3, 0, 6, 0, 0, 4, 7]
    c = rr.api.compose3x4(rr.transform3x4((1, 1, 1), (1, 0, 0)), t)
    assert list(c) == [2, 0, 0, 6, 0, 3, 0, 6, 0, 0, 4, 7]


def test_tile_index_math():
    W, H, tile, world = 100, 70, 32, 3
    D = rr.distributed
    counts = D.tile_counts(W, H, tile, world)
    assert sum(counts) == 4 * 3 and counts == [4, 4, 4]
    seen = np.zeros(W * H, dtype=np.int32)
    for r in range(world):
        n, idx = D.tile_pixel_index(W, H, tile, r, world)
        assert n == counts[r] * tile * tile
        valid = idx[idx >= 0]
        seen[valid] += 1
        assert (D.owner_map(W, H, tile, world).reshape(-1)[valid] == r).all()
    assert (seen == 1).all(), "every pixel is owned by exactly one rank"
    img = np.random.default_rng(0).random((H, W, 4), dtype=np.float32)
    out = np.zeros_like(img)
    for r in range(world):
        D.unpack_tiles_host(out, D.pack_tiles_host(img, tile, r, world), tile, r, world)
    assert np.array_equal(out, img)


