"""-m gpu, at BASELINE.json's full frame size (1920x1080, the config-2 / config-3 scenes): parity against
the oracle on a sample of framebuffer tiles (the oracle renders only those tiles - its tile partition
is the multi-GPU one), full-frame reservoir parity, and size-independent properties of the frame."""
import numpy as np
import pytest

import oracle_api as oa
import rust_renderer_amd as rr
from util import L2_TOL, per_pixel_l2, reference_density, run_frames

pytestmark = pytest.mark.gpu
W, H, TILE = 1920, 1080, 64


@pytest.fixture(scope="module")
def sponza():
    return rr.scenes.scene_for_config(1, tex_size=64)


@pytest.fixture(scope="module")
def sponza_lights():
    return rr.scenes.scene_for_config(2, tex_size=64)


def sampled(world, rank):
    return rr.distributed.owner_map(W, H, TILE, world) == rank


def test_sampled_tiles_match_oracle_config2(sponza):
    gpu = sponza.upload(rr.Renderer(W, H))
    cpu = sponza.upload(oa.OracleRenderer(W, H))
    cpu.set_tile_partition(11, 53, TILE)  # 9-10 of the 510 tiles, spread over the frame
    for r in (gpu, cpu):
        run_frames(r, sponza, W, H, 2, rr.PASS_REFERENCE_PT)
    mask = sampled(53, 11)
    assert mask.sum() >= 30_000
    a, b = gpu.read_accumulation()[mask], cpu.read_accumulation()[mask]
    assert per_pixel_l2(a, b) <= L2_TOL
    assert np.abs(a - b).max() <= 1e-4  # only the sky integral (device exp/pow vs libm) differs


@pytest.mark.parametrize("config, pass_mask, world, rank", [(1, rr.PASS_REFERENCE_PT, 53, 11), (2, rr.PASS_ALL, 47, 5)])
def test_timed_workload_tex1024_matches_oracle(config, pass_mask, world, rank):
    """the scene bench.py TIMES - configs[1] / configs[2] with the 1024^2 albedo maps (8x8-tiled texel layout on the device),
    1920x1080 - through the batched uh_render_frames path the timed region uses, against the oracle on sampled tiles.
    (bench.py repeats the comparison on the whole frame beside its timing: the `parity` block of its JSON line.)"""
    scene = rr.scenes.scene_for_config(config, tex_size=1024)
    gpu = scene.upload(rr.Renderer(W, H))
    cpu = scene.upload(oa.OracleRenderer(W, H))
    cpu.set_tile_partition(rank, world, TILE)
    frames = 3
    rr.FrameLoop(gpu, scene.make_view(W, H)).frames(frames, pass_mask)
    run_frames(cpu, scene, W, H, frames, pass_mask)
    mask = sampled(world, rank)
    a, b = gpu.read_accumulation()[mask], cpu.read_accumulation()[mask]
    assert per_pixel_l2(a / frames, b / frames) <= L2_TOL
    assert np.abs(a - b).max() <= 1e-4 * frames  # only the sky integral (device exp/pow vs libm) differs
    if config == 2:  # reservoir chain of the last frame: full frame, bit for bit
        for which in range(3):
            g, c = gpu.read_reservoirs(which), cpu.read_reservoirs(which)
            assert np.array_equal(g.view(np.uint32), c.view(np.uint32)), which


def test_restir_frame_matches_oracle_config3(sponza_lights):
    gpu = sponza_lights.upload(rr.Renderer(W, H))
    cpu = sponza_lights.upload(oa.OracleRenderer(W, H))
    cpu.set_tile_partition(5, 47, TILE)
    for r in (gpu, cpu):
        run_frames(r, sponza_lights, W, H, 2, rr.PASS_ALL)
    # G-buffer and all three reservoir buffers: full frame, bit for bit
    assert np.array_equal(gpu.read_gbuffer_position().view(np.uint32), cpu.read_gbuffer_position().view(np.uint32))
    for which in range(3):
        g, c = gpu.read_reservoirs(which), cpu.read_reservoirs(which)
        for f in ("Y", "M"):
            assert np.array_equal(g[f], c[f]), (which, f)
        for f in ("W_sum", "W_X"):
            assert np.array_equal(g[f].view(np.uint32), c[f].view(np.uint32)), (which, f)
    mask = sampled(47, 5)
    a, b = gpu.read_accumulation()[mask], cpu.read_accumulation()[mask]
    assert per_pixel_l2(a, b) <= L2_TOL
    # both halves of the reference's split screen (x > W/2 reads reservoirs) are in the sample
    xs = np.nonzero(mask)[1]
    assert (xs > W // 2).any() and (xs < W // 2).any()


def test_frame_invariants_and_determinism(sponza):
    frames = 3
    out = []
    for _ in range(2):
        r = sponza.upload(rr.Renderer(W, H))
        run_frames(r, sponza, W, H, frames, rr.PASS_REFERENCE_PT)
        out.append((r.read_accumulation(), r.get_stats()))
    (a, s), (b, t) = out
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "two runs of the same frames must agree bit for bit"
    assert list(s.rays) == list(t.rays)
    assert s.rays[0] == W * H * frames                       # one primary ray per pixel and sample
    assert s.closest_hits + s.misses == s.rays[0] + s.rays[1]  # every path ray ends in exactly one shader
    # a sun ray leaves every scattered hit (rgen:63-79), a bounce ray every scattered hit but the last bounce's
    assert s.rays[1] <= s.rays[2] <= s.closest_hits
    assert s.rays[3] == 0                                    # no lights in config 2
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all() and (a[..., 3] == 0).all()


def test_batched_in_flight_frames_equal_frame_by_frame(sponza):
    ref = sponza.upload(rr.Renderer(W, H))
    ref.set_option("frames_in_flight", 1)
    run_frames(ref, sponza, W, H, 7, rr.PASS_REFERENCE_PT)
    alt = sponza.upload(rr.Renderer(W, H))
    alt.set_option("batch_frames", 3)
    rr.FrameLoop(alt, sponza.make_view(W, H)).frames(7, rr.PASS_REFERENCE_PT)
    assert np.array_equal(ref.read_accumulation().view(np.uint32), alt.read_accumulation().view(np.uint32))
    assert list(ref.get_stats().rays) == list(alt.get_stats().rays)


def test_gpu_group_composes_the_full_frame(sponza):
    single = sponza.upload(rr.Renderer(W, H))
    group = sponza.upload(rr.MultiGpuRenderer(W, H, devices=[0, 0, 0, 0], tile_size=TILE))
    for r in (single, group):
        rr.FrameLoop(r, sponza.make_view(W, H)).frames(4, rr.PASS_REFERENCE_PT)
    assert np.array_equal(single.read_accumulation().view(np.uint32), group.read_accumulation().view(np.uint32))
    assert np.array_equal(single.read_output_bgra8(), group.read_output_bgra8())
    assert list(single.get_stats().rays) == list(group.get_stats().rays)


def test_moved_instances_refit_equals_rebuild(sponza):
    n = sponza.num_meshes
    moves = [(n - 1, rr.transform3x4((1, 1, 1), (3.0, 0.5, 1.0))), (n - 2, rr.transform3x4((1.3, 0.7, 1.0), (-4.0, 1.5, -1.0)))]
    refit = sponza.upload(rr.Renderer(W, H))
    for mesh, w in moves:
        refit.set_instance_transform(mesh, w)
    refit.rebuild_tlas()
    built = rr.Renderer(W, H)
    for model, transform in sponza.models:
        built.add_model(model, transform)
    for mesh, w in moves:
        built.set_instance_transform(mesh, w)
    built.initialize_raytracing()
    for r in (refit, built):
        run_frames(r, sponza, W, H, 2, rr.PASS_REFERENCE_PT)
    assert np.array_equal(refit.read_accumulation().view(np.uint32), built.read_accumulation().view(np.uint32))
    assert list(refit.get_stats().rays) == list(built.get_stats().rays)


def test_config4_scene_class_at_4k_sampled_tiles_and_group():
    """BASELINE configs[3]: the Bistro-class scene (2.88 M triangles, all four material types, 64 lights)
    at 3840x2160 - the largest frame and scene the configs name"""
    W4, H4 = 3840, 2160
    scene = rr.scenes.scene_for_config(3, tex_size=64)
    gpu = scene.upload(rr.Renderer(W4, H4))
    cpu = scene.upload(oa.OracleRenderer(W4, H4))
    cpu.set_tile_partition(7, 211, TILE)  # 9-10 of the 2040 tiles
    for r in (gpu, cpu):
        run_frames(r, scene, W4, H4, 2, rr.PASS_REFERENCE_PT, use_ris_light_sampling=0)
    mask = rr.distributed.owner_map(W4, H4, TILE, 211) == 7
    assert mask.sum() >= 30_000
    a, b = gpu.read_accumulation()[mask], cpu.read_accumulation()[mask]
    assert per_pixel_l2(a, b) <= L2_TOL
    s = gpu.get_stats()
    assert s.rays[0] == 2 * W4 * H4 and s.rays[3] > 0 and s.closest_hits + s.misses == s.rays[0] + s.rays[1]
    group = scene.upload(rr.MultiGpuRenderer(W4, H4, devices=[0, 0], tile_size=TILE))
    run_frames(group, scene, W4, H4, 2, rr.PASS_REFERENCE_PT, use_ris_light_sampling=0)
    assert np.array_equal(gpu.read_accumulation().view(np.uint32), group.read_accumulation().view(np.uint32))


def test_config0_rtiow_256_matches_oracle():
    """BASELINE configs[0]: RTIOW 3 spheres (Lambertian ground + centre, dielectric, metal; icospheres of 5
    subdivisions, 81,920 triangles) + 1 point light, 256x256, 1 spp, 5 bounces, sky + sun + uniform light
    sampling - the whole frame through the HIP path against the whole frame from the oracle"""
    scene = rr.scenes.scene_for_config(0)
    assert scene.num_triangles == 4 * 20 * 4 ** 5 and len(scene.lights) == 1
    w = h = 256
    gpu = scene.upload(rr.Renderer(w, h))
    cpu = scene.upload(oa.OracleRenderer(w, h))
    for r in (gpu, cpu):
        run_frames(r, scene, w, h, 1, rr.PASS_REFERENCE_PT)
    a, b = gpu.read_accumulation(), cpu.read_accumulation()
    assert per_pixel_l2(a, b) <= L2_TOL
    assert np.abs(a - b).max() <= 1e-4  # geometry, materials and light terms are bit-identical; the sky integral is not
    s, t = gpu.get_stats(), cpu.get_stats()
    assert list(s.rays) == list(t.rays) and s.rays[0] == w * h and s.rays[3] > 0  # light shadow rays were traced
    assert s.closest_hits == t.closest_hits and s.misses == t.misses
    # all three sphere materials are in the image: the 8-bit outputs agree to 1 LSB (sRGB pow)
    assert np.abs(gpu.read_output_bgra8().astype(np.int16) - cpu.read_output_bgra8().astype(np.int16)).max() <= 1


def test_config4_isosurface_512_at_1080p_sampled_tiles():
    """BASELINE configs[4]: the 512^3 iso-surface of the reference's marching-cubes density field
    (marching_cubes.comp:83-103) extracted on the GPU, path traced at 1920x1080. The oracle gets the very
    triangles the device extracted (read_mesh) and renders a sample of tiles spread over the frame."""
    scene = rr.scenes.scene_for_config(4)
    gpu = rr.Renderer(W, H)
    iso, ntri = gpu.add_isosurface_mesh(512, 0.0, 32.0)
    assert iso == 0 and 500_000 < ntri < 3_000_000  # "~1-3 M tris" (marching cubes: about half of what marching tetrahedra cut)
    for model, transform in scene.models:  # the ground plane
        gpu.add_model(model, transform)
    gpu.initialize_raytracing()
    v, idx = gpu.read_mesh(iso)
    assert len(idx) == 3 * ntri
    # every extracted vertex lies on the iso-surface to a fraction of a cell (sampled: 5 M vertices)
    pick = np.arange(0, len(v), 97)
    assert np.abs(reference_density(v["pos"][pick, :3].astype(np.float64))).max() < 0.2 * 32.0 / 512
    cpu = oa.OracleRenderer(W, H)
    cpu.add_mesh(v, idx, rr.make_material(base_color=(0.8, 0.8, 0.8, 1.0), diffuse_map=cpu.default_diffuse_map()), None)
    for model, transform in scene.models:
        cpu.add_model(model, transform)
    cpu.initialize_raytracing()
    cpu.set_tile_partition(3, 61, TILE)  # 8-9 of the 510 tiles
    for r in (gpu, cpu):
        run_frames(r, scene, W, H, 2, rr.PASS_REFERENCE_PT)
    mask = sampled(61, 3)
    assert mask.sum() >= 30_000
    a, b = gpu.read_accumulation()[mask], cpu.read_accumulation()[mask]
    assert per_pixel_l2(a, b) <= L2_TOL
    assert np.abs(a - b).max() <= 1e-4
    s = gpu.get_stats()
    assert s.rays[0] == 2 * W * H and s.closest_hits + s.misses == s.rays[0] + s.rays[1] and s.rays[3] == 0
    acc = gpu.read_accumulation()
    assert np.isfinite(acc).all() and (acc[..., :3] >= 0).all()
    # the surface is in the picture: the frame's centre rows hit geometry, not only sky
    assert s.closest_hits > W * H
