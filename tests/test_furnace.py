"""The reference's one verification aid: FURNACE_TEST (reference.rmiss:14-28 - the miss shader returns white). With every
material's albedo exactly 1 (white map, base colour 1: rchit:40-41), sun and lights off, a sample's radiance is
throughput x 1 = EXACTLY 1.0 when its path leaves the scene (rgen:48-57) and exactly 0.0 when it is still inside after
view.num_bounces hits - whatever the geometry, the BVH, the RNG or the traversal order. So after N frames every pixel of the
accumulation image is an integer in [0, N], the image sums to the number of miss-shader invocations, and in a scene every
path leaves within the bounce budget every pixel is exactly N. A known answer with no oracle in the loop: the HIP path is
held against the arithmetic identity itself (the oracle gets the same test, on the CPU)."""
import copy

import numpy as np
import pytest

import rust_renderer_amd as rr
from rust_renderer_amd.types import LAMBERTIAN


def albedo_one(scene, retype=True):
    """the same geometry with albedo-1 Lambertian materials: default white map, base colour (1, 1, 1, 1)"""
    s = copy.deepcopy(scene)
    for model, _ in s.models:
        for m in model.meshes:
            m.base_color = (1.0, 1.0, 1.0, 1.0)
            m.texture = None
            if retype:
                m.material_type, m.material_property = LAMBERTIAN, 0.0
    s.lights = []
    s.view_flags = dict(s.view_flags, sky_enabled=1, sun_shadow_enabled=0, lights_enabled=0, use_ris_light_sampling=0, num_bounces=64, samples_per_frame=1)
    return s


def furnace_frames(renderer, scene, W, H, frames, batched=False):
    renderer.set_option("furnace", 1)
    loop = rr.FrameLoop(renderer, scene.make_view(W, H))
    if batched:
        loop.frames(frames, rr.PASS_REFERENCE_PT)
    else:
        for _ in range(frames):
            loop.frame(rr.PASS_REFERENCE_PT)
    return renderer.read_accumulation()[..., :3], renderer.get_stats()


def check_identity(acc, stats, frames, all_escape):
    assert np.isfinite(acc).all()
    assert np.array_equal(acc, np.rint(acc)), "a furnace sample is exactly 0 or 1: the accumulation must hold integers"
    assert acc.min() >= 0.0 and acc.max() <= float(frames)
    assert np.array_equal(acc[..., 0], acc[..., 1]) and np.array_equal(acc[..., 0], acc[..., 2])
    # all-Lambertian scene: a path ends only in the miss shader, so the image sums to the miss count
    assert int(acc[..., 0].astype(np.float64).sum()) == int(stats.misses)
    if all_escape:
        assert (acc == float(frames)).all(), f"{int((acc[..., 0] != frames).sum())} pixels are not exactly {frames}"
    else:
        # an interior keeps some paths bouncing for all 64 hits (Cornell box: ~1 %, the atrium of the timed workload: a fifth of the
        # pixels have one such sample in four): every sample is still exactly 0 or 1, and most are 1
        assert acc[..., 0].astype(np.float64).mean() > 0.8 * frames


def open_scene():
    """open surfaces only - a tessellated floor and a back wall meeting in an L: a path that hits one of them scatters back into
    the half-space it came from (rchit:47-50 after the flip of rchit:35-37), so it can only ever meet the OTHER surface next, and
    a ray that slips through the crack between two triangles of a surface (Moeller-Trumbore is not watertight) is simply outside.
    Nothing can trap a path: every sample is out within a handful of bounces and is exactly 1. (Closed meshes do trap the odd
    path that slips through a crack into their interior - 12 of 393 k samples on two convex bodies - which is why the closed
    scenes below are held to "integers that sum to the miss count" instead.)"""
    from rust_renderer_amd.scenes import Mesh, Model, Scene, quad
    from rust_renderer_amd.camera import Camera

    meshes = [Mesh(*quad((-3, 0, -2), (0, 0, 4), (6, 0, 0), 12, 8), LAMBERTIAN, 0.0, (0.3, 0.6, 0.9, 1.0), None, name="floor"),
              Mesh(*quad((-3, 0, -2), (6, 0, 0), (0, 3, 0), 12, 6), LAMBERTIAN, 0.0, (0.9, 0.2, 0.1, 1.0), None, name="wall")]
    cam = Camera((0.5, 1.6, 4.0), (0.0, 0.6, 0.0), 60.0, 1.0, 0.01, 1000.0)
    return Scene("furnace-open", [(Model(meshes, []), None)], [], cam, dict(sky_enabled=1))


def closed_bodies_scene():
    """two convex closed meshes far apart: almost every path is out after one or two hits"""
    from rust_renderer_amd.scenes import Mesh, Model, Scene, box, icosphere
    from rust_renderer_amd.api import transform3x4
    from rust_renderer_amd.camera import Camera

    sv, si = icosphere(3)
    meshes = [Mesh(sv, si, LAMBERTIAN, 0.0, (0.3, 0.6, 0.9, 1.0), None, transform3x4((0.6,) * 3, (-0.9, 0.0, 0.0))),
              Mesh(*box((0, 0, 0), (1, 1, 1), 2), LAMBERTIAN, 0.0, (0.9, 0.2, 0.1, 1.0), None, transform3x4((0.4, 0.5, 0.3), (0.9, 0.1, 0.2)))]
    cam = Camera((0.0, 0.4, 3.0), (0.0, 0.0, 0.0), 60.0, 1.0, 0.01, 1000.0)
    return Scene("furnace-bodies", [(Model(meshes, []), None)], [], cam, dict(sky_enabled=1))


def test_oracle_furnace_is_exactly_one(oa):
    W = H = 48
    scene = albedo_one(open_scene())
    acc, st = furnace_frames(scene.upload(oa.OracleRenderer(W, H)), scene, W, H, 3)
    check_identity(acc, st, 3, all_escape=True)
    # ... and the flag is what does it: with sky_enabled = 0 and no furnace the same frames are black (rmiss:26-27)
    off = scene.upload(oa.OracleRenderer(W, H))
    loop = rr.FrameLoop(off, scene.make_view(W, H, sky_enabled=0))
    loop.frame(rr.PASS_REFERENCE_PT)
    assert not off.read_accumulation()[..., :3].any()


def test_oracle_furnace_ignores_sky_enabled(oa):
    W = H = 32
    scene = albedo_one(open_scene())
    scene.view_flags["sky_enabled"] = 0  # rmiss:14: the whole sky block is compiled out under FURNACE_TEST
    acc, st = furnace_frames(scene.upload(oa.OracleRenderer(W, H)), scene, W, H, 2)
    check_identity(acc, st, 2, all_escape=True)


def test_oracle_furnace_cornell_is_integer_valued(oa):
    W = H = 40
    scene = albedo_one(rr.scenes.cornell_scene(subdivisions=1, tex_size=8))
    acc, st = furnace_frames(scene.upload(oa.OracleRenderer(W, H)), scene, W, H, 2)
    check_identity(acc, st, 2, all_escape=False)


@pytest.mark.gpu
@pytest.mark.parametrize("batched", [False, True])
def test_hip_furnace_open_scene_every_pixel_exactly_one(batched):
    W, H, N = 256, 192, 8
    scene = albedo_one(open_scene())
    acc, st = furnace_frames(scene.upload(rr.Renderer(W, H, device=0)), scene, W, H, N, batched)
    check_identity(acc, st, N, all_escape=True)


@pytest.mark.gpu
def test_hip_furnace_closed_bodies():
    W, H, N = 256, 192, 8
    scene = albedo_one(closed_bodies_scene())
    acc, st = furnace_frames(scene.upload(rr.Renderer(W, H, device=0)), scene, W, H, N, True)
    check_identity(acc, st, N, all_escape=False)
    assert (acc[..., 0] == float(N)).mean() > 0.999


@pytest.mark.gpu
def test_hip_furnace_cornell():
    W, H, N = 160, 160, 6
    scene = albedo_one(rr.scenes.cornell_scene(subdivisions=2, tex_size=16))
    acc, st = furnace_frames(scene.upload(rr.Renderer(W, H, device=0)), scene, W, H, N, True)
    check_identity(acc, st, N, all_escape=False)


@pytest.mark.gpu
def test_hip_furnace_sponza_class_scene():
    """the timed workload's geometry (262 k triangles, 103 meshes) at a reduced frame: integers everywhere, image sum = misses"""
    W, H, N = 480, 270, 4
    scene = albedo_one(rr.scenes.scene_for_config(1, tex_size=64))
    acc, st = furnace_frames(scene.upload(rr.Renderer(W, H, device=0)), scene, W, H, N, True)
    check_identity(acc, st, N, all_escape=False)


@pytest.mark.gpu
def test_hip_furnace_with_the_reference_materials_is_still_integer_valued():
    """metal and dielectric set colour = 1 (rchit:58,82), a DiffuseLight ends the path with colour 1 (rchit:85-89): with the Cornell
    scene's own material TYPES at albedo 1 a sample is still exactly 0 or 1 (the image sum then counts unscattered hits too)"""
    W, H, N = 128, 128, 4
    scene = albedo_one(rr.scenes.cornell_scene(subdivisions=2, tex_size=16), retype=False)
    acc, st = furnace_frames(scene.upload(rr.Renderer(W, H, device=0)), scene, W, H, N, True)
    assert np.array_equal(acc, np.rint(acc)) and acc.min() >= 0 and acc.max() <= N
    assert int(acc[..., 0].astype(np.float64).sum()) >= int(st.misses)
