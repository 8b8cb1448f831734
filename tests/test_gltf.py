"""CPU: glTF ingestion on the reference's own self-contained assets
(read from the reference checkout where it is mounted; skipped elsewhere - nothing here runs on the
GPU box), then the reference's Cornell scene script (prototype/src/scenes.rs:58-100, without the
FlightHelmet whose textures are absent) rendered by the oracle."""
import os

import numpy as np
import pytest

import oracle_api as oa
import rust_renderer_amd as rr
from rust_renderer_amd import gltf

REF = "/root/reference/prototype/data/models"
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "sphere.gltf")), reason="reference assets not mounted")


@needs_ref
def test_sphere_gltf_counts_and_geometry():
    m = gltf.load_gltf(os.path.join(REF, "sphere.gltf"))
    assert len(m.meshes) == 1 and m.meshes[0].num_triangles == 4512
    v = m.meshes[0].vertices
    assert np.allclose(np.linalg.norm(v["pos"][:, :3], axis=1), 1.0, atol=2e-3)
    assert np.allclose(np.linalg.norm(v["normal"][:, :3], axis=1), 1.0, atol=1e-3)
    assert (v["pos"][:, 3] == 0).all() and (v["color"] == 1).all() and (v["tangent"] == 0).all()
    assert m.meshes[0].indices.max() < len(v)
    assert np.array_equal(m.meshes[0].transform, rr.identity3x4())


@needs_ref
def test_cornell_gltf_matches_the_asset_inventory():
    m = gltf.load_gltf(os.path.join(REF, "CornellBox-Original.gltf"))
    assert len(m.meshes) == 8 and sum(x.num_triangles for x in m.meshes) == 32 and not m.textures
    names = [x.name for x in m.meshes]
    assert names == ["floor", "ceiling", "backWall", "rightWall", "leftWall", "shortBox", "tallBox", "Light"]
    floor = m.meshes[0]
    assert np.allclose(floor.base_color, (0.725, 0.71, 0.68, 1.0), atol=1e-6)
    # the node carries a +90 degree rotation about x (quaternion (0.7071, 0, 0, 0.7071)): y -> z
    t = floor.transform.reshape(3, 4)
    assert np.allclose(t[:, :3], [[1, 0, 0], [0, 0, -1], [0, 1, 0]], atol=1e-6) and np.allclose(t[:, 3], 0)
    assert np.allclose(gltf.instance_transform_3x4(t), floor.transform, atol=1e-6), "a pure rotation survives the SRT round trip"


def test_cube_mirrors_model_loader():
    c = gltf.load_cube().meshes[0]
    assert len(c.vertices) == 24 and c.num_triangles == 12
    tri = c.vertices["pos"][c.indices.reshape(-1, 3)][:, :, :3]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0)  # twelve unit-square halves
    assert np.allclose(np.abs(c.vertices["pos"][:, :3]), 0.5)


def test_srt_round_trip_drops_shear():
    m = np.array([[2, 0.5, 0, 1], [0, 3, 0, 2], [0, 0, 4, 3]], dtype=np.float32)
    out = gltf.instance_transform_3x4(m).reshape(3, 4)
    cols = out[:, :3]
    assert np.allclose(cols.T @ cols, np.diag(np.linalg.norm(m[:, :3], axis=0) ** 2), atol=1e-4), "recomposed axes are orthogonal"
    assert np.allclose(out[:, 3], [1, 2, 3])


@needs_ref
def test_reference_cornell_scene_script_renders():
    cornell = gltf.load_gltf(os.path.join(REF, "CornellBox-Original.gltf"))
    light = gltf.load_cube()
    light.meshes[0].material_type = rr.DIFFUSE_LIGHT  # scenes.rs:79-80
    cam = rr.camera.Camera((0.0, 0.9, 2.0), (0.0, 0.5, 0.0), 60.0, 1.0, 0.01, 1000.0)  # scenes.rs:63-66
    scene = rr.scenes.Scene("reference_cornell", [(cornell, None), (light, rr.transform3x4((0.50, 0.05, 0.35), (0.0, 1.95, 0.0)))], [], cam,
                            dict(lights_enabled=0, sun_shadow_enabled=0, sky_enabled=1))
    W = H = 64
    results = []
    for brute in (False, True):
        o = scene.upload(oa.OracleRenderer(W, H, brute_force=brute))
        loop = rr.FrameLoop(o, scene.make_view(W, H))
        for _ in range(4):
            loop.frame(rr.PASS_REFERENCE_PT)
        results.append(o.read_accumulation())
        stats = o.get_stats()
    assert np.array_equal(results[0].view(np.uint32), results[1].view(np.uint32))
    img = results[0][..., :3] / 4
    assert np.isfinite(img).all() and img.max() <= 1.0 + 1e-5
    assert stats.rays[0] == 4 * W * H and stats.closest_hits > 0.9 * stats.rays[0], "the camera sits inside the box"
    left, right = img[H // 2, 4], img[H // 2, W - 5]
    assert left[0] > left[1] * 1.5 and right[1] > right[0] * 1.5, "red wall on the left, green wall on the right"
