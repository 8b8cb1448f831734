"""The reference's own self-contained assets through the whole path (SURVEY.md section 8f row N1, section 8c "self-contained
reference assets usable as fixtures"): prototype/data/models/CornellBox-Original.gltf and sphere.gltf, committed as DATA in
tests/golden/reference_assets.npz (generator: tests/golden/make_reference_asset_fixtures.py), placed by the reference's scene
scripts restated here:

  create_scene                prototype/src/scenes.rs:16-24   10 point lights at ((i / 30) * 20, 3.5, (i % 30) * 20)
  create_cornell_box_scene    prototype/src/scenes.rs:58-100  camera (0, 0.9, 2) -> (0, 0.5, 0); the box at the origin; load_cube as a
                              DiffuseLight scaled (0.50, 0.05, 0.35) at (0, 1.95, 0)  [the FlightHelmet of :73-76,94-98 needs texture
                              files that are absent from the mount and is left out]
  create_sponza_scene         prototype/src/scenes.rs:102-150 camera (-10.28, 2.10, -0.18) -> (0, 0.5, 0); sphere.gltf twice, size 0.6:
                              Metal at (-3, 2.65, 0.7), Dielectric 1.5 at (-3, 0.65, 0.7)  [Sponza.bin is absent from the mount]

CPU: the fixture equals what the loader reads from the reference checkout (where it is mounted), and the asset inventory of
SURVEY.md section 2 row 22. GPU: the HIP path against the oracle on both scenes, every pass of the graph."""
import os

import numpy as np
import pytest

import oracle_api as oa
import rust_renderer_amd as rr
from rust_renderer_amd import gltf
from rust_renderer_amd.scenes import Mesh, Model, Scene
from rust_renderer_amd.types import VERTEX_DTYPE
from util import L2_TOL, per_pixel_l2

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


@pytest.fixture(scope="module")
def assets():
    return np.load(os.path.join(HERE, "golden", "reference_assets.npz"))


def model_from(assets, key):
    meshes = []
    for i in range(int(assets[f"{key}_count"])):
        v = np.ascontiguousarray(assets[f"{key}_{i}_vertices"]).view(VERTEX_DTYPE).reshape(-1)
        meshes.append(Mesh(v, assets[f"{key}_{i}_indices"], rr.LAMBERTIAN, 0.0, tuple(float(x) for x in assets[f"{key}_{i}_base_color"]), None,
                           assets[f"{key}_{i}_transform"].copy(), name=str(assets[f"{key}_{i}_name"])))
    return Model(meshes, [])


REFERENCE_LIGHTS = [(float((i // 30) * 20), 3.5, float((i % 30) * 20)) for i in range(10)]  # scenes.rs:16-24


def reference_cornell_scene(assets):
    light = gltf.load_cube()
    light.meshes[0].material_type = rr.DIFFUSE_LIGHT  # scenes.rs:79-80
    cam = rr.camera.Camera((0.0, 0.9, 2.0), (0.0, 0.5, 0.0), 60.0, 1.0, 0.01, 1000.0)  # scenes.rs:63-66, fov / near / far of main.rs:44-52
    return Scene("reference_cornell", [(model_from(assets, "cornell"), None), (light, rr.transform3x4((0.50, 0.05, 0.35), (0.0, 1.95, 0.0)))],
                 REFERENCE_LIGHTS, cam, dict(sky_enabled=1, sun_shadow_enabled=1, lights_enabled=1, use_ris_light_sampling=1))


def reference_spheres_scene(assets):
    metal, glass = model_from(assets, "sphere"), model_from(assets, "sphere")
    metal.meshes[0].material_type = rr.METAL                                       # scenes.rs:116-117
    glass.meshes[0].material_type, glass.meshes[0].material_property = rr.DIELECTRIC, 1.5  # scenes.rs:118-122
    cam = rr.camera.Camera((-10.28, 2.10, -0.18), (0.0, 0.5, 0.0), 60.0, 1.0, 0.01, 1000.0)  # scenes.rs:107-110
    place = lambda y: rr.transform3x4((0.6, 0.6, 0.6), (-3.0, y, 0.7))             # scenes.rs:130-149
    return Scene("reference_spheres", [(metal, place(2.65)), (glass, place(0.65))], REFERENCE_LIGHTS, cam,
                 dict(sky_enabled=1, sun_shadow_enabled=1, lights_enabled=1, use_ris_light_sampling=0))


def test_fixture_matches_the_asset_inventory(assets):
    cornell, sphere = model_from(assets, "cornell"), model_from(assets, "sphere")
    assert [m.name for m in cornell.meshes] == ["floor", "ceiling", "backWall", "rightWall", "leftWall", "shortBox", "tallBox", "Light"]
    assert sum(m.num_triangles for m in cornell.meshes) == 32 and len(sphere.meshes) == 1 and sphere.meshes[0].num_triangles == 4512
    assert np.allclose(cornell.meshes[0].base_color, (0.725, 0.71, 0.68, 1.0), atol=1e-6)
    v = sphere.meshes[0].vertices
    assert np.allclose(np.linalg.norm(v["pos"][:, :3], axis=1), 1.0, atol=2e-3) and (v["color"] == 1).all()
    # Renderer::initialize's default maps (renderer.rs:202-220): constant images, RGB8 expanded to RGBA8 with alpha 255
    assert tuple(assets["default_white_texture_shape"]) == (128, 128) and (assets["default_white_texture"] == 255).all()
    assert np.array_equal(assets["default_default_metallic_roughness_unique"], [[0, 255, 0, 255]])
    assert (np.abs(assets["default_flat_normal_map_unique"].astype(int) - [127, 127, 255, 255]) <= 2).all()  # a dithered (127.5, 127.5, 255) image


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "prototype/data/models/sphere.gltf")), reason="reference checkout not mounted")
def test_loader_reproduces_the_fixture_from_the_reference_files(assets):
    """pins rust-renderer_amd/gltf.py + image_decode.py against the files themselves"""
    for key, path in (("cornell", "CornellBox-Original.gltf"), ("sphere", "sphere.gltf")):
        live, fix = gltf.load_gltf(os.path.join(REF, "prototype/data/models", path)), model_from(assets, key)
        assert len(live.meshes) == len(fix.meshes)
        for a, b in zip(live.meshes, fix.meshes):
            assert a.vertices.tobytes() == b.vertices.tobytes() and np.array_equal(a.indices, b.indices)
            assert np.array_equal(np.asarray(a.transform, dtype=np.float32), b.transform) and a.name == b.name
    from rust_renderer_amd import image_decode
    white = image_decode.load_image_rgba8(open(os.path.join(REF, "utopian/data/textures/defaults/white_texture.png"), "rb").read())
    assert white.shape == (128, 128, 4) and np.array_equal(white[:16, :16], assets["default_white_texture"])


def test_default_texture_maps_follow_renderer_initialize(assets):
    """DEFAULT_TEXTURE_MAP fallbacks of Renderer::add_model (renderer.rs:222-262) on the oracle backend (no GPU needed)"""
    o = oa.OracleRenderer(16, 16)
    d = o.initialize([np.ascontiguousarray(assets[f"default_{n}"]) for n in ("white_texture", "flat_normal_map", "white_texture", "default_metallic_roughness")])
    assert (d["diffuse"], d["normal"], d["occlusion"], d["metallic_roughness"]) == (0, 1, 2, 3)  # the first four bindless textures
    assert o.default_diffuse_map() == 0
    ids = o.add_model(model_from(assets, "cornell"))
    assert len(ids) == 8


def run(renderer, scene, W, H, frames, mask, **view):
    loop = rr.FrameLoop(renderer, scene.make_view(W, H, **view))
    for _ in range(frames):
        loop.frame(mask)
    return renderer


def test_oracle_renders_the_reference_cornell_script(assets):
    scene = reference_cornell_scene(assets)
    W = H = 48
    a = run(scene.upload(oa.OracleRenderer(W, H)), scene, W, H, 3, rr.PASS_ALL)
    b = run(scene.upload(oa.OracleRenderer(W, H, brute_force=True)), scene, W, H, 3, rr.PASS_ALL)
    assert np.array_equal(a.read_accumulation().view(np.uint32), b.read_accumulation().view(np.uint32))
    img = a.read_accumulation()[..., :3] / 3
    assert np.isfinite(img).all()
    left, right = img[H // 2, 3], img[H // 2, W - 4]
    assert left[0] > left[1] * 1.5 and right[1] > right[0] * 1.5, "red wall on the left, green wall on the right"


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["cornell", "spheres"])
def test_hip_path_matches_oracle_on_the_reference_assets(assets, which):
    """every pass of build_path_tracing_render_graph (gbuffer, reset, initial RIS, temporal, spatial, reference_pt) on the
    reference's own geometry: G-buffer, reservoirs and ray counts bit for bit, radiance within the per-pixel tolerance
    (only the sky integral differs), and bit for bit once the sky is off"""
    scene = reference_cornell_scene(assets) if which == "cornell" else reference_spheres_scene(assets)
    W, H = 160, 120
    gpu = run(scene.upload(rr.Renderer(W, H)), scene, W, H, 4, rr.PASS_ALL)
    cpu = run(scene.upload(oa.OracleRenderer(W, H)), scene, W, H, 4, rr.PASS_ALL)
    assert np.array_equal(gpu.read_gbuffer_position().view(np.uint32), cpu.read_gbuffer_position().view(np.uint32))
    for r in range(3):
        g, c = gpu.read_reservoirs(r), cpu.read_reservoirs(r)
        assert np.array_equal(g["Y"], c["Y"]) and np.array_equal(g["M"], c["M"])
        assert np.array_equal(g["W_sum"].view(np.uint32), c["W_sum"].view(np.uint32)) and np.array_equal(g["W_X"].view(np.uint32), c["W_X"].view(np.uint32))
    assert list(gpu.get_stats().rays) == list(cpu.get_stats().rays)
    assert per_pixel_l2(gpu.read_accumulation(), cpu.read_accumulation()) <= L2_TOL
    assert np.abs(gpu.read_output_bgra8().astype(np.int16) - cpu.read_output_bgra8().astype(np.int16)).max() <= 1
    gpu2 = run(scene.upload(rr.Renderer(W, H)), scene, W, H, 3, rr.PASS_ALL, sky_enabled=0)
    cpu2 = run(scene.upload(oa.OracleRenderer(W, H)), scene, W, H, 3, rr.PASS_ALL, sky_enabled=0)
    assert np.array_equal(gpu2.read_accumulation().view(np.uint32), cpu2.read_accumulation().view(np.uint32))
    assert gpu2.read_accumulation()[..., :3].max() > 0.0
