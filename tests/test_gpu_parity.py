"""-m gpu: the HIP path (called through the C ABI) against the CPU oracle on the same seeded
inputs. Integer results (hit ids, ray counts, reservoir Y / M) are compared bit-exactly; linear
radiance within the per-pixel L2 tolerance BASELINE.json states (1e-3); the 8-bit output image
within 1 LSB (sRGB pow differs by an ulp between libm and the device library)."""
import numpy as np
import pytest

import oracle_api as oa
import rust_renderer_amd as rr
from util import L2_TOL, extract_isosurface, make_pair, per_pixel_l2, random_rays, reference_density, run_frames, torture_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cornell():
    return rr.scenes.cornell_scene(subdivisions=2, tex_size=16)


@pytest.fixture(scope="module")
def atrium():
    return rr.scenes.sponza_class_scene(detail=0.12, tex_size=32, with_spheres=True, num_lights=64, sphere_subdivisions=2)


def test_trace_closest_bit_exact(atrium):
    gpu, cpu = make_pair(atrium, 8, 8)
    rays = random_rays(((-15, 0.2, -7), (15, 9, 7)), 200_000, seed=11)
    tg, mg, pg = gpu.trace_closest(rays)
    tc, mc, pc = cpu.trace_closest(rays)
    assert np.array_equal(mg, mc) and np.array_equal(pg, pc)
    assert np.array_equal(tg.view(np.uint32), tc.view(np.uint32)), "t/u/v must match the oracle bit for bit"
    assert (mg != 0xFFFFFFFF).mean() > 0.5


def test_trace_closest_matches_brute_force(cornell):
    gpu, cpu = make_pair(cornell, 8, 8, brute_force=True)
    rays = random_rays(((-0.9, 0.1, -0.9), (0.9, 1.9, 0.9)), 20_000, seed=5)
    tg, mg, pg = gpu.trace_closest(rays)
    tc, mc, pc = cpu.trace_closest(rays)
    assert np.array_equal(mg, mc) and np.array_equal(pg, pc)
    assert np.array_equal(tg.view(np.uint32), tc.view(np.uint32))


def test_trace_any_matches_closest(atrium):
    gpu, cpu = make_pair(atrium, 8, 8)
    rays = random_rays(((-15, 0.2, -7), (15, 9, 7)), 100_000, seed=3, tmax=6.0)
    occ = gpu.trace_any(rays)
    _, mc, _ = cpu.trace_closest(rays)
    assert np.array_equal(occ.astype(bool), mc != 0xFFFFFFFF)


def test_axis_aligned_and_degenerate_rays(atrium):
    gpu, cpu = make_pair(atrium, 8, 8)
    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, 0.9863939, 0.1643990], [1, 1, 0], [0, 1e-20, 1]], dtype=np.float32)
    base = random_rays(((-15, 0.2, -7), (15, 9, 7)), 4096, seed=17)
    rays = np.repeat(base, len(dirs), axis=0)
    rays[:, 4:7] = np.tile(dirs, (len(base), 1))
    tg, mg, pg = gpu.trace_closest(rays)
    tc, mc, pc = cpu.trace_closest(rays)
    assert np.array_equal(mg, mc) and np.array_equal(pg, pc)
    assert np.array_equal(tg.view(np.uint32), tc.view(np.uint32))


@pytest.mark.parametrize("frames,spp", [(1, 1), (3, 1), (2, 3)])
def test_path_trace_matches_oracle(cornell, frames, spp):
    W, H = 96, 80
    gpu, cpu = make_pair(cornell, W, H)
    for r in (gpu, cpu):
        run_frames(r, cornell, W, H, frames, rr.PASS_REFERENCE_PT, samples_per_frame=spp)
    total = frames * spp
    a, b = gpu.read_accumulation(), cpu.read_accumulation()
    assert np.isfinite(a).all()
    l2 = per_pixel_l2(a / total, b / total)
    assert l2 <= L2_TOL, f"per-pixel L2 {l2}"
    sg, sc = gpu.get_stats(), cpu.get_stats()
    assert list(sg.rays)[:4] == list(sc.rays)[:4], "ray counts by kind must equal the oracle's"
    assert sg.closest_hits == sc.closest_hits and sg.misses == sc.misses
    og, oc = gpu.read_output_bgra8().astype(np.int32), cpu.read_output_bgra8().astype(np.int32)
    assert np.abs(og - oc).max() <= 1
    assert (og[..., 3] == 0).all()


@pytest.mark.parametrize("sun", [(0.3, 0.8, 0.2), (-0.3, 0.8, 0.2), (0.3, -0.8, 0.2), (0.3, 0.8, -0.2), (-0.3, -0.8, 0.2), (-0.3, 0.8, -0.2), (0.3, -0.8, -0.2),
                                 (-0.3, -0.8, -0.2), (0.0, 1.0, 0.0), (-0.0, 1.0, -0.0), (0.0, -1.0, 1e-35)])
def test_sun_direction_octants_match_oracle(atrium, sun):
    """the node step picks its near / far plane words by the sign of 1 / direction: every octant of the sun's shadow rays, and
    zero / negative-zero / tiny components, against the oracle with the sky off"""
    W, H = 96, 54
    gpu, cpu = make_pair(atrium, W, H)
    for r in (gpu, cpu):
        loop = rr.FrameLoop(r, atrium.make_view(W, H, sky_enabled=0, lights_enabled=0, sun_shadow_enabled=1))
        loop.view.sun_dir[:] = list(sun)
        for _ in range(2):
            loop.frame(rr.PASS_REFERENCE_PT)
    assert np.array_equal(gpu.read_accumulation().view(np.uint32), cpu.read_accumulation().view(np.uint32))
    assert list(gpu.get_stats().rays)[:4] == list(cpu.get_stats().rays)[:4]
    assert gpu.read_accumulation()[..., :3].max() > 0 or sun[1] < 0, "a sun above the horizon lights something"


def test_path_trace_atrium_all_materials(atrium):
    W, H = 160, 90
    gpu, cpu = make_pair(atrium, W, H)
    for r in (gpu, cpu):
        run_frames(r, atrium, W, H, 2, rr.PASS_REFERENCE_PT, use_ris_light_sampling=0)
    a, b = gpu.read_accumulation(), cpu.read_accumulation()
    l2 = per_pixel_l2(a / 2, b / 2)
    assert l2 <= L2_TOL, f"per-pixel L2 {l2}"
    assert list(gpu.get_stats().rays)[:4] == list(cpu.get_stats().rays)[:4]


def test_gbuffer_and_restir_chain_bit_exact(atrium):
    W, H = 128, 72
    gpu, cpu = make_pair(atrium, W, H)
    for r in (gpu, cpu):
        run_frames(r, atrium, W, H, 3, rr.PASS_RESTIR)
    gg, gc = gpu.read_gbuffer_position(), cpu.read_gbuffer_position()
    assert np.array_equal(gg.view(np.uint32), gc.view(np.uint32)), "G-buffer positions must match bit for bit"
    for which in range(3):
        rg, rc = gpu.read_reservoirs(which), cpu.read_reservoirs(which)
        assert np.array_equal(rg["Y"], rc["Y"]) and np.array_equal(rg["M"], rc["M"]), f"reservoir {which} Y/M"
        assert np.array_equal(rg["W_sum"].view(np.uint32), rc["W_sum"].view(np.uint32)), f"reservoir {which} W_sum"
        assert np.array_equal(rg["W_X"].view(np.uint32), rc["W_X"].view(np.uint32)), f"reservoir {which} W_X"
    assert (gpu.read_reservoirs(2)["M"] > 1).any(), "temporal history must build up over frames"


def test_full_frame_with_restir(atrium):
    W, H = 128, 72
    gpu, cpu = make_pair(atrium, W, H)
    for r in (gpu, cpu):
        run_frames(r, atrium, W, H, 3, rr.PASS_ALL)
    a, b = gpu.read_accumulation(), cpu.read_accumulation()
    l2 = per_pixel_l2(a / 3, b / 3)
    assert l2 <= L2_TOL, f"per-pixel L2 {l2}"
    assert list(gpu.get_stats().rays) == list(cpu.get_stats().rays)


def test_restir_flags_off_copy_through(atrium):
    W, H = 64, 36
    gpu, cpu = make_pair(atrium, W, H)
    for r in (gpu, cpu):
        run_frames(r, atrium, W, H, 2, rr.PASS_RESTIR, temporal_reuse_enabled=0, spatial_reuse_enabled=0)
    r0, r2 = gpu.read_reservoirs(0), gpu.read_reservoirs(2)
    assert np.array_equal(r0.view(np.uint8), r2.view(np.uint8))
    assert np.array_equal(r2.view(np.uint8), cpu.read_reservoirs(2).view(np.uint8))


@pytest.mark.parametrize("W,H,tile", [(96, 80, 16), (200, 120, 32)])  # the second: 7 x 4 tiles, last column and row partial, 10 / 9 / 9 per rank
def test_tile_partition_composes_to_full_frame(cornell, W, H, tile):
    """the device side of the composition with three contexts standing for three ranks: uh_pack_tiles -> device-to-device copies (what
    the grouped ncclSend / ncclRecv of uh_rccl_gather_tiles deliver) -> uh_unpack_tiles + uh_resolve_output, and the one-launch form
    uh_compose_tiles the in-library gather ends with, on another root"""
    from util import DeviceBuffer

    world, frames = 3, 2
    full = cornell.upload(rr.Renderer(W, H))
    run_frames(full, cornell, W, H, frames, rr.PASS_REFERENCE_PT)
    ref = full.read_accumulation()
    root = None
    parts, rays = [], 0
    for rank in range(world):
        r = cornell.upload(rr.Renderer(W, H))
        r.set_tile_partition(rank, world, tile)
        run_frames(r, cornell, W, H, frames, rr.PASS_REFERENCE_PT)
        rays += r.get_stats().path_rays
        n = r.tile_pack_count(rank)
        buf = DeviceBuffer(n * 16)
        r.pack_tiles(buf.ptr, n)
        parts.append((rank, buf, n))
        if rank == 0:
            root = r
    assert rays == full.get_stats().path_rays, "the ranks' ray counts add up to the full frame's"
    for rank, buf, n in parts[1:]:
        root.unpack_tiles(rank, buf.ptr, n)
    # the same composition in one launch (uh_compose_tiles: what the root of uh_rccl_gather_tiles runs behind the receives), on rank 1 as the root
    stride = max(n for _, _, n in parts)
    every = DeviceBuffer(world * stride * 16)
    for rank, buf, n in parts:
        if rank != 1:  # the root's own slot is not read: left zeroed
            every.copy_from(buf, n * 16, dst_offset=rank * stride * 16)
    other = cornell.upload(rr.Renderer(W, H))
    other.set_tile_partition(1, world, tile)
    run_frames(other, cornell, W, H, frames, rr.PASS_REFERENCE_PT)
    other.compose_tiles(every.ptr, stride, frames)
    root.resolve_output(frames)
    for r in (root, other):
        assert np.array_equal(r.read_accumulation().view(np.uint32), ref.view(np.uint32)), "tile-partitioned frame must be bit-identical"
        assert np.array_equal(r.read_output_bgra8(), full.read_output_bgra8())
    with pytest.raises(rr.UtopianError):
        other.compose_tiles(every.ptr, 16, frames)  # stride smaller than a rank's tiles


def test_error_paths():
    r = rr.Renderer(16, 16)
    with pytest.raises(rr.UtopianError, match="NOT_BUILT"):
        r.render_frame(rr.scenes.cornell_scene(1, 4).make_view(16, 16))
    with pytest.raises(rr.UtopianError):
        r.set_instance_transform(5, rr.identity3x4())
    with pytest.raises(rr.UtopianError):
        r.set_option("no_such_option", 1)


def test_empty_scene_renders_sky():
    W, H = 32, 32
    scene = rr.scenes.Scene("empty", [], [], rr.camera.Camera((0, 1, 0), (0, 1, -1), 60.0, 1.0), dict(lights_enabled=0))
    gpu, cpu = make_pair(scene, W, H)
    for r in (gpu, cpu):
        run_frames(r, scene, W, H, 1, rr.PASS_REFERENCE_PT)
    a, b = gpu.read_accumulation(), cpu.read_accumulation()
    assert per_pixel_l2(a, b) <= L2_TOL
    assert a[..., :3].max() > 0.05
    assert gpu.get_stats().rays[0] == W * H and gpu.get_stats().rays[1] == 0


def test_instance_transform_rebuild(cornell):
    W, H = 64, 64
    gpu, cpu = make_pair(cornell, W, H)
    for r in (gpu, cpu):
        r.set_instance_transform(6, rr.transform3x4((0.25, 0.35, 0.25), (-0.3, 0.5, 0.1)))
        r.initialize_raytracing()
        run_frames(r, cornell, W, H, 1, rr.PASS_REFERENCE_PT)
    assert per_pixel_l2(gpu.read_accumulation(), cpu.read_accumulation()) <= L2_TOL


@pytest.mark.parametrize("batch,in_flight,spp,limit", [(3, 2, 1, 999999), (8, 3, 1, 999999), (4, 1, 2, 999999), (3, 3, 1, 4)])
def test_batched_frames_equal_frame_by_frame(cornell, batch, in_flight, spp, limit):
    """uh_render_frames (several frames per wavefront, several wavefronts in flight) must equal the
    frame-by-frame protocol bit for bit, including the accumulation-limit freeze and partial batches"""
    W, H, frames = 96, 80, 7
    ref = cornell.upload(rr.Renderer(W, H))
    ref.set_option("frames_in_flight", 1)
    loop = run_frames(ref, cornell, W, H, frames, rr.PASS_REFERENCE_PT, samples_per_frame=spp, accumulation_limit=limit)
    alt = cornell.upload(rr.Renderer(W, H))
    alt.set_option("batch_frames", batch)
    alt.set_option("frames_in_flight", in_flight)
    loop2 = rr.FrameLoop(alt, cornell.make_view(W, H, samples_per_frame=spp, accumulation_limit=limit))
    loop2.frames(frames, rr.PASS_REFERENCE_PT)
    assert loop2.view.total_samples == loop.view.total_samples == frames * spp
    assert np.array_equal(ref.read_accumulation().view(np.uint32), alt.read_accumulation().view(np.uint32))
    assert np.array_equal(ref.read_output_bgra8(), alt.read_output_bgra8())
    assert list(ref.get_stats().rays) == list(alt.get_stats().rays)
    # and continuing frame by frame after a batched run keeps matching
    loop.frame(rr.PASS_REFERENCE_PT)
    loop2.frame(rr.PASS_REFERENCE_PT)
    assert np.array_equal(ref.read_accumulation().view(np.uint32), alt.read_accumulation().view(np.uint32))


def test_batched_frames_with_tile_partition(cornell):
    W, H, frames = 96, 80, 5
    ref = cornell.upload(rr.Renderer(W, H))
    ref.set_tile_partition(1, 3, 16)
    run_frames(ref, cornell, W, H, frames, rr.PASS_REFERENCE_PT)
    alt = cornell.upload(rr.Renderer(W, H))
    alt.set_tile_partition(1, 3, 16)
    alt.set_option("batch_frames", 4)
    rr.FrameLoop(alt, cornell.make_view(W, H)).frames(frames, rr.PASS_REFERENCE_PT)
    assert np.array_equal(ref.read_accumulation().view(np.uint32), alt.read_accumulation().view(np.uint32))


@pytest.mark.parametrize("batch,in_flight,frames,split", [(4, 2, 9, 0), (3, 3, 11, 1), (0, 4, 14, 0), (2, 1, 5, 1)])
def test_batched_restir_frames_equal_frame_by_frame(atrium, batch, in_flight, frames, split):
    """reservoir passes + path tracer of a static camera through uh_render_frames: B reservoir chains back to back (a ring of
    spatial buffers), then one wavefront whose paths sample from their own frame's spatial reservoirs - bit for bit the
    frame-by-frame protocol: radiance, all three reservoir buffers, ray counts; also when continued either way"""
    W, H = 96, 54
    kw = dict(use_ris_light_sampling=1)
    ref = atrium.upload(rr.Renderer(W, H))
    ref.set_option("frames_in_flight", 1)
    ref.set_option("full_frame_restir", split)
    loop = run_frames(ref, atrium, W, H, frames, rr.PASS_ALL, **kw)
    alt = atrium.upload(rr.Renderer(W, H))
    alt.set_option("batch_frames", batch)
    alt.set_option("frames_in_flight", in_flight)
    alt.set_option("full_frame_restir", split)
    loop2 = rr.FrameLoop(alt, atrium.make_view(W, H, **kw))
    loop2.frames(frames, rr.PASS_ALL)
    assert loop2.view.total_samples == loop.view.total_samples == frames
    assert (ref.read_reservoirs(2)["M"] > 1).any()
    def same():
        assert np.array_equal(ref.read_accumulation().view(np.uint32), alt.read_accumulation().view(np.uint32))
        for which in range(3):
            assert np.array_equal(ref.read_reservoirs(which), alt.read_reservoirs(which)), which
        assert list(ref.get_stats().rays) == list(alt.get_stats().rays)
    same()
    loop.frame(rr.PASS_ALL)
    loop2.frame(rr.PASS_ALL)
    same()
    for _ in range(3):
        loop.frame(rr.PASS_ALL)
    loop2.frames(3, rr.PASS_ALL)
    same()


def test_reservoir_passes_alone_run_frame_by_frame(atrium):
    """uh_render_frames without the path-tracing pass: nothing to batch, the chains run one after the other"""
    W, H = 64, 36
    mask = rr.PASS_ALL & ~rr.PASS_REFERENCE_PT
    ref = atrium.upload(rr.Renderer(W, H))
    loop = run_frames(ref, atrium, W, H, 4, mask)
    alt = atrium.upload(rr.Renderer(W, H))
    alt.set_option("batch_frames", 4)
    loop2 = rr.FrameLoop(alt, atrium.make_view(W, H))
    for k in range(2):  # two calls of two frames: the camera is at rest, so prev_frame_projection_view only changes after the first
        v = loop2.view
        v.num_lights = alt.get_num_lights()
        v.total_samples += 1
        if k == 0:
            alt.render_frames(v, mask, 1)
            loop2.end_frame()
            v.total_samples += 1
            alt.render_frames(v, mask, 1)
        else:
            alt.render_frames(v, mask, 2)
            v.total_samples += 1
    assert loop2.view.total_samples == loop.view.total_samples == 4
    for which in range(3):
        assert np.array_equal(ref.read_reservoirs(which), alt.read_reservoirs(which)), which


@pytest.mark.parametrize("W,H,world,tile,batch", [(97, 61, 1, 64, 3), (97, 61, 3, 20, 4), (65, 3, 2, 7, 2), (1, 1, 1, 64, 2), (130, 70, 8, 64, 8)])
def test_awkward_sizes_match_oracle(cornell, W, H, world, tile, batch):
    """widths that are not multiples of the 64-path run, tiles that do not divide the frame, 1x1:
    every rank's pixels equal the oracle's full frame bit for bit (no sky in view => exact)"""
    frames = 3
    cpu = cornell.upload(oa.OracleRenderer(W, H))
    run_frames(cpu, cornell, W, H, frames, rr.PASS_REFERENCE_PT, sky_enabled=0)
    ref = cpu.read_accumulation()
    owner = rr.distributed.owner_map(W, H, tile, world)
    total_rays = 0
    for rank in range(world):
        gpu = cornell.upload(rr.Renderer(W, H))
        if world > 1:
            gpu.set_tile_partition(rank, world, tile)
        gpu.set_option("batch_frames", batch)
        rr.FrameLoop(gpu, cornell.make_view(W, H, sky_enabled=0)).frames(frames, rr.PASS_REFERENCE_PT)
        acc = gpu.read_accumulation()
        mine = owner == rank
        assert np.array_equal(acc[mine].view(np.uint32), ref[mine].view(np.uint32)), f"rank {rank}"
        assert (acc[~mine] == 0).all()
        total_rays += gpu.get_stats().path_rays
    assert total_rays == cpu.get_stats().path_rays


@pytest.mark.parametrize("overrides", [dict(num_bounces=0), dict(samples_per_frame=0), dict(num_bounces=1), dict(num_bounces=17),
                                       dict(lights_enabled=1, num_lights=0), dict(sun_shadow_enabled=0, lights_enabled=0)])
def test_degenerate_view_settings_match_oracle(cornell, overrides):
    W, H = 64, 48
    gpu, cpu = make_pair(cornell, W, H)
    ov = dict(overrides)
    if ov.pop("num_lights", None) == 0:
        # a scene without lights: lights_enabled with zero lights samples index 0 of an empty table
        scene = rr.scenes.Scene("nolights", cornell.models, [], cornell.camera, dict(cornell.view_flags))
        gpu, cpu = make_pair(scene, W, H)
    else:
        scene = cornell
    for r in (gpu, cpu):
        run_frames(r, scene, W, H, 2, rr.PASS_REFERENCE_PT, **ov)
    a, b = gpu.read_accumulation(), cpu.read_accumulation()
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert per_pixel_l2(np.nan_to_num(a), np.nan_to_num(b)) <= L2_TOL
    assert list(gpu.get_stats().rays)[:4] == list(cpu.get_stats().rays)[:4]


# ---- on-device refit (uh_refit_acceleration; raytracing.rs:400-459 rebuild_tlas) ---------------------
def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    mx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    my = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    mz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (mz @ my @ mx).astype(np.float32)


def _moved(scene, renderer_factory, W, H, moves, how):
    """how = 'build': transforms set before the one and only build; 'refit': built at the scene's
    own transforms, then moved and refitted on the device; 'flag': as refit, but requested through
    view.rebuild_tlas like the application does."""
    r = renderer_factory()
    if how == "build":
        for model, transform in scene.models:
            r.add_model(model, transform)
        for p in scene.lights:
            r.add_light(p, (1.0, 1.0, 1.0), 1.0)
        for mesh, w in moves:
            r.set_instance_transform(mesh, w)
        r.initialize_raytracing()
    else:
        scene.upload(r)
        run_frames(r, scene, W, H, 1, rr.PASS_REFERENCE_PT)  # frames in flight on the old tree
        for mesh, w in moves:
            r.set_instance_transform(mesh, w)
        if how == "refit":
            r.rebuild_tlas()
        r.reset_accumulation()
        r.reset_stats()
    return r


@pytest.mark.parametrize("how", ["refit", "flag"])
def test_refit_equals_rebuild_bit_for_bit(atrium, how):
    W, H = 96, 64
    n_mesh = atrium.num_meshes
    moves = [(n_mesh - 1, rr.transform3x4((0.8, 1.1, 0.9), (1.5, 0.7, -0.4), _rot(0.3, 1.0, -0.2))),
             (n_mesh - 2, rr.transform3x4((1.0, 1.0, 1.0), (-2.0, 1.2, 0.6))),
             (3, rr.transform3x4((1.02, 0.97, 1.0), (0.05, 0.0, -0.03), _rot(0.0, 0.02, 0.0)))]
    built = _moved(atrium, lambda: rr.Renderer(W, H), W, H, moves, "build")
    refit = _moved(atrium, lambda: rr.Renderer(W, H), W, H, moves, how)
    rays = random_rays(((-14, -1, -7), (14, 12, 7)), 20000, seed=77)
    overrides = dict(rebuild_tlas=1) if how == "flag" else {}
    for r in (built, refit):
        run_frames(r, atrium, W, H, 2, rr.PASS_ALL, **overrides)
    a, b = built.trace_closest(rays), refit.trace_closest(rays)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert np.array_equal(built.trace_any(rays), refit.trace_any(rays))
    assert np.array_equal(built.read_accumulation().view(np.uint32), refit.read_accumulation().view(np.uint32))
    assert np.array_equal(built.read_gbuffer_position().view(np.uint32), refit.read_gbuffer_position().view(np.uint32))
    assert list(built.get_stats().rays) == list(refit.get_stats().rays)


def test_refit_matches_oracle(cornell):
    W, H = 64, 48
    moves = [(6, rr.transform3x4((0.25, 0.35, 0.25), (-0.3, 0.5, 0.1), _rot(0.0, 0.6, 0.0)))]
    gpu = _moved(cornell, lambda: rr.Renderer(W, H), W, H, moves, "refit")
    cpu = _moved(cornell, lambda: oa.OracleRenderer(W, H, threads=3), W, H, moves, "refit")
    for r in (gpu, cpu):
        run_frames(r, cornell, W, H, 2, rr.PASS_ALL)
    assert per_pixel_l2(gpu.read_accumulation(), cpu.read_accumulation()) <= L2_TOL
    assert list(gpu.get_stats().rays) == list(cpu.get_stats().rays)


def test_refit_needs_a_built_tree(cornell):
    r = rr.Renderer(32, 32)
    with pytest.raises(rr.UtopianError):
        r.rebuild_tlas()
    cornell.upload(r)
    r.rebuild_tlas()  # nothing moved: a refit of the tree onto itself
    m = cornell.models[0][0].meshes[0]
    r.add_mesh(m.vertices, m.indices, m.material_struct(), None)
    with pytest.raises(rr.UtopianError):
        r.rebuild_tlas()
    v = cornell.make_view(32, 32, rebuild_tlas=1)
    with pytest.raises(rr.UtopianError):
        r.render_frame(v, rr.PASS_REFERENCE_PT)


# ---- uh_mgpu_*: several GPUs behind one process (here: several contexts on the one GPU) ---------------
@pytest.mark.parametrize("ngpus,tile", [(3, 16), (2, 64), (5, 8)])
def test_gpu_group_equals_single_context(cornell, ngpus, tile):
    W, H = 80, 56
    single = cornell.upload(rr.Renderer(W, H))
    group = cornell.upload(rr.MultiGpuRenderer(W, H, devices=[0] * ngpus, tile_size=tile))
    assert group.get_num_lights() == single.get_num_lights()
    for r in (single, group):
        loop = run_frames(r, cornell, W, H, 2, rr.PASS_ALL)       # G-buffer + ReSTIR + path tracing, frame by frame
        loop.view.use_ris_light_sampling = 0
        loop.frames(5, rr.PASS_REFERENCE_PT)                       # batched progressive frames
    assert np.array_equal(single.read_accumulation().view(np.uint32), group.read_accumulation().view(np.uint32))
    assert np.array_equal(single.read_output_bgra8(), group.read_output_bgra8())
    a, b = single.get_stats(), group.get_stats()
    assert list(a.rays) == list(b.rays) and a.closest_hits == b.closest_hits and a.misses == b.misses
    # a moved instance reaches every GPU; the group refits on all of them
    moved = rr.transform3x4((0.25, 0.35, 0.25), (-0.3, 0.5, 0.1))
    for r in (single, group):
        r.set_instance_transform(6, moved)
        r.rebuild_tlas()
        r.reset_accumulation()
        run_frames(r, cornell, W, H, 2, rr.PASS_REFERENCE_PT)
    assert np.array_equal(single.read_accumulation().view(np.uint32), group.read_accumulation().view(np.uint32))


def test_gpu_group_errors():
    with pytest.raises(rr.UtopianError):
        rr.MultiGpuRenderer(32, 32, devices=[0, 4096])
    g = rr.MultiGpuRenderer(32, 32, devices=[0, 0])
    with pytest.raises(rr.UtopianError):
        g.render_frame(rr.types.ViewUniformData(), rr.PASS_REFERENCE_PT)  # nothing built


def test_group_composition_is_ordered_by_events(atrium):
    """uh_mgpu_compose with the copy LATE BY CONSTRUCTION (round 4's soak saw stale tiles in 2 % of its scenes, only with other
    processes on the GPU): GPUs 1 and 2 are slowed down (one block per CU, one frame in flight) and asked for 24 frames, the
    composition is enqueued at once - nothing has been waited for -, so when GPU 0's share is done the other two are still tracing:
    their packs, the peer copies behind them and GPU 0's one-launch composition are held in order by events alone. Then more frames
    and another composition behind the first (the staging buffer is reused), again without a wait."""
    import ctypes as C

    W, H, tile = 256, 144, 32
    single = atrium.upload(rr.Renderer(W, H))
    group = atrium.upload(rr.MultiGpuRenderer(W, H, devices=[0, 0, 0], tile_size=tile))
    lib = rr.load_library()
    lib.uh_set_option.argtypes, lib.uh_set_option.restype = [C.c_void_p, C.c_char_p, C.c_int], C.c_int
    for i in (1, 2):
        ctx = lib.uh_mgpu_context(group._ctx, i)
        for name, value in ((b"trace_blocks_per_cu", 1), (b"frames_in_flight", 1), (b"batch_frames", 2)):
            assert lib.uh_set_option(ctx, name, value) == 0
    loops = {r: rr.FrameLoop(r, atrium.make_view(W, H, lights_enabled=0)) for r in (single, group)}
    for r in (single, group):
        loops[r].frames(24, rr.PASS_REFERENCE_PT)
    group.compose()  # enqueued behind 24 frames still in flight on GPUs 1 and 2
    want24 = single.read_accumulation()
    got24 = group.read_accumulation()  # (composed already: reads GPU 0)
    assert np.array_equal(got24.view(np.uint32), want24.view(np.uint32)), "stale tiles: the composition ran ahead of a peer copy"
    assert np.array_equal(group.read_output_bgra8(), single.read_output_bgra8())
    for r in (single, group):
        loops[r].frames(6, rr.PASS_REFERENCE_PT)
    group.compose()
    for r in (single, group):
        loops[r].frames(2, rr.PASS_REFERENCE_PT)  # frames enqueued BEHIND a composition accumulate behind it
    assert np.array_equal(group.read_accumulation().view(np.uint32), single.read_accumulation().view(np.uint32))
    assert np.array_equal(group.read_output_bgra8(), single.read_output_bgra8())
    assert list(group.get_stats().rays) == list(single.get_stats().rays)


# ---- geometric torture: the intersection contract (min t, ties -> smaller mesh<<22|prim) -------------
def test_intersection_contract_on_torture_geometry():
    scene = torture_scene()
    gpu, cpu = make_pair(scene, 8, 8, brute_force=True)
    rng = np.random.default_rng(5)
    n = 60_000
    o = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.5, 1.5, n), rng.uniform(3, 6, n)], 1)
    tgt = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-1.2, 1.2, n), np.zeros(n)], 1)
    # a third of the rays aim exactly at vertices / edges / the shared diagonal
    special = np.array([[-1, -1, 0], [1, 1, 0], [0, 0, 0], [1, -1, 0], [0.5, 0.5, 0], [0, 0, 2], [0.25, 0.25, 0.5], [1, 0, 0]], dtype=np.float64)
    tgt[::3] = special[rng.integers(0, len(special), len(tgt[::3]))]
    o[::6, :2] = tgt[::6, :2]  # and some of those straight down the z axis
    rays = np.empty((n, 8), dtype=np.float32)
    rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = o, 0.001, tgt - o, 10000.0
    tg, mg, pg = gpu.trace_closest(rays)
    tc, mc, pc = cpu.trace_closest(rays)
    assert np.array_equal(mg, mc) and np.array_equal(pg, pc)
    assert np.array_equal(tg.view(np.uint32), tc.view(np.uint32))
    hit0 = mg == 0
    assert hit0.sum() > 1000 and not (mg == 1).any() and not (mg == 2).any(), "coincident surfaces: the smaller key must win every tie"
    assert not (mg == 3).any(), "zero-area triangles are never hit (det == 0)"
    assert np.array_equal(gpu.trace_any(rays).astype(bool), mc != 0xFFFFFFFF)
    # the same contract through the whole path tracer
    for r in (gpu, cpu):
        run_frames(r, scene, 8, 8, 2, rr.PASS_ALL)
    assert per_pixel_l2(gpu.read_accumulation(), cpu.read_accumulation()) <= L2_TOL


# ---- ReSTIR frames in flight: per-pass hazards, double-buffered spatial reservoirs ---------------------
@pytest.mark.parametrize("pass_by_pass", [False, True])
def test_restir_frames_in_flight_equal_serial(atrium, pass_by_pass):
    """frame f+1's G-buffer / reservoir passes overlap frame f's path tracer; the result must equal the
    strictly serial schedule bit for bit, also when the six passes are issued as six calls (the C++ graph)"""
    W, H, frames = 480, 270, 8
    out = []
    for in_flight in (1, 3):
        r = atrium.upload(rr.Renderer(W, H))
        r.set_option("frames_in_flight", in_flight)
        loop = rr.FrameLoop(r, atrium.make_view(W, H))
        for f in range(frames):
            if pass_by_pass and in_flight == 3:
                v = loop.view
                v.num_lights = r.get_num_lights()
                v.total_samples += v.samples_per_frame
                for bit in (rr.PASS_GBUFFER, rr.PASS_RESET_RESERVOIRS, rr.PASS_INITIAL_RIS, rr.PASS_TEMPORAL_REUSE, rr.PASS_SPATIAL_REUSE, rr.PASS_REFERENCE_PT):
                    r.render_frame(v, bit)
                loop.end_frame()
            else:
                loop.frame(rr.PASS_ALL)
            if f == 4:  # a read-back in the middle must see a consistent state and not disturb the rest
                mid = r.read_reservoirs(2)
                assert mid["M"].max() > 0
        out.append((r.read_accumulation(), [r.read_reservoirs(k) for k in range(3)], r.read_gbuffer_position(), r.get_stats()))
    (a, ra, ga, sa), (b, rb, gb, sb) = out
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32))
    for x, y in zip(ra, rb):
        assert x.tobytes() == y.tobytes()
    assert list(sa.rays) == list(sb.rays)


# ---- randomised scenes: triangle soups with every material type, random instance transforms and lights ---
def _soup_scene(seed):
    from rust_renderer_amd.camera import Camera
    from rust_renderer_amd.scenes import Mesh, Model, Scene, _pack_vertices

    rng = np.random.default_rng(seed)
    meshes, textures = [], []
    for t in range(3):
        textures.append(rng.integers(0, 256, size=(8 << t, 4 << t, 4), dtype=np.uint8))
    for m in range(int(rng.integers(3, 9))):
        nt = int(rng.integers(1, 200))
        centre = rng.uniform(-2, 2, size=(nt, 1, 3))
        pos = (centre + rng.normal(scale=rng.uniform(0.05, 0.8), size=(nt, 3, 3))).reshape(-1, 3).astype(np.float32)
        nrm = rng.normal(size=(nt * 3, 3)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        uv = rng.uniform(-1.5, 2.5, size=(nt * 3, 2)).astype(np.float32)
        mtype = int(rng.integers(0, 5))  # 4 = the Cook-Torrance extension
        prop = float(rng.uniform(0.0, 0.6)) if mtype == 1 else float(rng.uniform(1.1, 2.0))
        mesh = Mesh(_pack_vertices(pos, nrm, uv), np.arange(nt * 3, dtype=np.uint32), mtype, prop,
                    tuple(float(x) for x in rng.uniform(0.2, 1.0, size=3)) + (1.0,), int(rng.integers(0, 3)) if rng.random() < 0.7 else None)
        mesh.metallic, mesh.roughness = float(rng.uniform(0, 1)), float(rng.uniform(0.05, 1))
        if rng.random() < 0.5:
            mesh.transform = rr.transform3x4(tuple(rng.uniform(0.3, 1.7, size=3) * rng.choice([-1, 1], size=3)), tuple(rng.uniform(-1, 1, size=3)), _rot(*rng.uniform(-3, 3, size=3)))
        meshes.append(mesh)
    lights = [tuple(float(x) for x in rng.uniform(-3, 3, size=3)) for _ in range(int(rng.integers(1, 40)))]
    cam = Camera(tuple(float(x) for x in rng.uniform(-1, 1, size=3) + np.array([0, 0, 6.0])), (0.0, 0.0, 0.0), float(rng.uniform(30, 100)), 1.0, 0.01, 1000.0)
    return Scene(f"soup{seed}", [(Model(meshes, textures), None)], lights, cam)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_soup_scenes_match_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    scene = _soup_scene(seed)
    W, H = int(rng.integers(17, 97)), int(rng.integers(9, 71))
    overrides = dict(samples_per_frame=int(rng.integers(1, 4)), num_bounces=int(rng.integers(1, 8)), sky_enabled=int(rng.integers(0, 2)),
                     sun_shadow_enabled=int(rng.integers(0, 2)), lights_enabled=int(rng.integers(0, 2)), use_ris_light_sampling=int(rng.integers(0, 2)),
                     temporal_reuse_enabled=int(rng.integers(0, 2)), spatial_reuse_enabled=int(rng.integers(0, 2)),
                     max_num_lights_used=int(rng.choice([1, 7, 10000])))
    gpu, cpu = make_pair(scene, W, H, threads=3)
    for r in (gpu, cpu):
        run_frames(r, scene, W, H, 3, rr.PASS_ALL, **overrides)
    assert per_pixel_l2(gpu.read_accumulation(), cpu.read_accumulation()) <= L2_TOL, overrides
    assert np.array_equal(gpu.read_gbuffer_position().view(np.uint32), cpu.read_gbuffer_position().view(np.uint32))
    for which in range(3):
        assert gpu.read_reservoirs(which).tobytes() == cpu.read_reservoirs(which).tobytes(), (which, overrides)
    assert list(gpu.get_stats().rays) == list(cpu.get_stats().rays), overrides
    if not overrides["sky_enabled"]:
        assert np.array_equal(gpu.read_accumulation().view(np.uint32), cpu.read_accumulation().view(np.uint32)), "without the sky integral the image is bit-exact"


# ---- on-device LBVH build (option "device_build"): same hits as the host SAH tree, bit for bit -----------
def _single_triangle_scene():
    from rust_renderer_amd.camera import Camera
    from rust_renderer_amd.scenes import Mesh, Model, Scene, _pack_vertices
    pos = np.float32([[-1, -1, 0], [1, -1, 0], [0, 1, 0]])
    m = Mesh(_pack_vertices(pos, np.tile(np.float32([0, 0, 1]), (3, 1)), np.zeros((3, 2), np.float32)), np.arange(3, dtype=np.uint32))
    return Scene("one", [(Model([m], []), None)], [(0.0, 0.0, 3.0)], Camera((0, 0, 4), (0, 0, 0), 60.0, 1.0, 0.01, 100.0))


@pytest.mark.parametrize("kind", [1, 2, (1, 0), (1, 24)])
@pytest.mark.parametrize("which", ["cornell", "atrium", "torture", "soup", "one", "empty"])
def test_device_build_equals_host_build(cornell, atrium, which, kind):
    """option device_build: 1 = PLOC with a host SAH tree over the last ploc_sah_top clusters (default 1024: on these small
    scenes the rounds never start and the whole binary tree is the SAH top; (1, 0) = PLOC to the root; (1, 24) = both halves),
    2 = radix tree (csrc/lbvh.hip)"""
    sah_top = None
    if isinstance(kind, tuple):
        kind, sah_top = kind
    from rust_renderer_amd.camera import Camera
    from rust_renderer_amd.scenes import Model, Scene
    scene = {"cornell": cornell, "atrium": atrium, "torture": torture_scene(), "soup": _soup_scene(11), "one": _single_triangle_scene(),
             "empty": Scene("empty", [(Model([], []), None)], [], Camera((0, 0, 4), (0, 0, 0), 60.0, 1.0, 0.01, 100.0))}[which]
    W, H = 72, 48
    host = scene.upload(rr.Renderer(W, H))
    dev = rr.Renderer(W, H)
    dev.set_option("device_build", kind)
    if sah_top is not None:
        dev.set_option("ploc_sah_top", sah_top)
    scene.upload(dev)
    assert dev.get_stats().bvh_triangles == host.get_stats().bvh_triangles == scene.num_triangles
    rays = random_rays(((-3, -1, -3), (3, 3, 3)) if which != "atrium" else ((-14, 0, -7), (14, 10, 7)), 30000, seed=21)
    if scene.num_triangles:
        for a, b in zip(host.trace_closest(rays), dev.trace_closest(rays)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(host.trace_any(rays), dev.trace_any(rays))
    else:
        # zero triangles (zero-size packet arrays, a root with no child): every query misses, on both builders
        for r in (host, dev):
            tuv, mesh, prim = r.trace_closest(rays[:500])
            assert (tuv[:, 0] == -1.0).all() and (mesh == 0xFFFFFFFF).all() and (prim == 0xFFFFFFFF).all()
            assert not r.trace_any(rays[:500]).any()
    for r in (host, dev):
        run_frames(r, scene, W, H, 2, rr.PASS_ALL if scene.lights else rr.PASS_REFERENCE_PT)
    assert np.array_equal(host.read_accumulation().view(np.uint32), dev.read_accumulation().view(np.uint32))
    assert list(host.get_stats().rays) == list(dev.get_stats().rays)
    if scene.num_meshes > 2:
        # moved instances: device rebuild, then a refit of the device-built tree, against the host rebuild
        w = rr.transform3x4((0.6, 0.8, 0.7), (0.2, 0.3, -0.1), _rot(0.1, 0.5, -0.2))
        for r in (host, dev):
            r.set_instance_transform(1, w)
            r.initialize_raytracing()
        w2 = rr.transform3x4((0.9, 0.9, 0.9), (-0.2, 0.1, 0.3))
        host.set_instance_transform(2, w2)
        host.initialize_raytracing()
        dev.set_instance_transform(2, w2)
        dev.rebuild_tlas()
        for a, b in zip(host.trace_closest(rays), dev.trace_closest(rays)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _chain_scene(clusters=300):
    """clusters on a geometric series towards the origin: a SAH (and a Morton) tree over them is a chain far deeper than the
    traversal stack (16 LDS + 96 scratch entries = 37 levels); the builders must notice and fall back to a balanced tree"""
    from rust_renderer_amd.camera import Camera
    from rust_renderer_amd.scenes import Mesh, Model, Scene, _pack_vertices
    u = rr.scenes.hash_floats(0xC4A1, clusters * 3 * 12).reshape(clusters, 3, 12)
    pos = []
    for k in range(clusters):
        s = np.float32(0.78) ** k * 4.0
        for i in range(3):
            c = np.float32([s * (1.0 + 0.1 * u[k, i, 0]), s * 0.3 * (u[k, i, 1] - 0.5), s * 0.3 * (u[k, i, 2] - 0.5)])
            pos += [c + s * 0.2 * (u[k, i, 3 + 3 * v: 6 + 3 * v] - 0.5) for v in range(3)]
    pos = np.asarray(pos, dtype=np.float32)
    m = Mesh(_pack_vertices(pos, np.tile(np.float32([0, 0, 1]), (len(pos), 1)), np.zeros((len(pos), 2), np.float32)), np.arange(len(pos), dtype=np.uint32))
    return Scene("chain", [(Model([m], []), None)], [], Camera((0, 0, 6), (0, 0, 0), 60.0, 1.0, 0.01, 100.0))


@pytest.mark.parametrize("device_build", [0, 1, 2])
def test_deep_clustered_geometry_keeps_every_hit(device_build):
    """a tree deeper than the traversal stack used to drop subtrees silently (trav_push past capacity). Now
    the host builder rebuilds balanced when the SAH tree has more than kMaxTreeLevels levels, and a device build of
    such a scene falls back to the host builder: closest hits and any-hits equal the oracle's brute force."""
    scene = _chain_scene()
    gpu = rr.Renderer(32, 32)
    gpu.set_option("device_build", device_build)
    scene.upload(gpu)
    cpu = scene.upload(oa.OracleRenderer(32, 32, brute_force=True))
    n = 20000
    u = rr.scenes.hash_floats(77, 6 * n).reshape(n, 6)
    rays = np.empty((n, 8), dtype=np.float32)
    rays[:, 0:3] = (u[:, :3] - 0.5) * np.float32([12, 4, 4])
    rays[:, 3] = 0.001
    # aim at points on the chain's axis, most of them close to the origin where the deep clusters are
    target = np.zeros((n, 3), dtype=np.float32)
    target[:, 0] = 4.4 * u[:, 3] ** 6
    rays[:, 4:7] = target - rays[:, 0:3] + (u[:, 3:6] - 0.5) * 0.05
    rays[:, 7] = 10000.0
    for a, b in zip(gpu.trace_closest(rays), cpu.trace_closest(rays)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert (gpu.trace_closest(rays)[0][:, 0] > 0).mean() > 0.05
    assert np.array_equal(gpu.trace_any(rays), cpu.trace_any(rays))


def test_non_finite_geometry_is_rejected_at_the_boundary(cornell):
    m = cornell.models[0][0].meshes[0]
    r = rr.Renderer(16, 16)
    for bad in (np.nan, np.inf):
        v = m.vertices.copy()
        v["pos"][1, 2] = bad
        with pytest.raises(rr.UtopianError):
            r.add_mesh(v, m.indices, m.material_struct(), None)
    mesh = r.add_mesh(m.vertices, m.indices, m.material_struct(), None)
    w = rr.identity3x4()
    w[3] = np.inf
    with pytest.raises(rr.UtopianError):
        r.set_instance_transform(mesh, w)
    # finite input whose world-space image overflows: the builders cope (such triangles are simply never hit)
    big = rr.transform3x4((3e38, 3e38, 3e38), (0, 0, 0))
    r.set_instance_transform(mesh, big)
    for dev in (0, 1):
        r.set_option("device_build", dev)
        r.initialize_raytracing()
        t, _, _ = r.trace_closest(random_rays(((-1, -1, -1), (1, 1, 1)), 1000, seed=3))
        assert np.isfinite(t[:, 0]).all()


# ---- GPU marching-cubes extraction of the reference's density field (uh_add_isosurface_mesh) -------------
def test_isosurface_extraction_is_marching_cubes_and_renders_like_the_oracle():
    res, lo, hi = 48, 0.0, 32.0
    W, H = 96, 64
    cell = (hi - lo) / res
    gpu = rr.Renderer(W, H)
    # this repository's own tables (round 3's form: watertight bit for bit, slivers dropped); the default - the reference's
    # triangles, cell by cell - is held against the oracle in tests/test_marching_cubes.py and renders below like any mesh
    gpu.set_option("iso_reference_triangulation", 0)
    mesh, ntri = gpu.add_isosurface_mesh(res, lo, hi)
    assert mesh is not None and ntri > 1000
    v, idx = gpu.read_mesh(mesh)
    assert len(idx) == 3 * ntri and np.array_equal(idx, np.arange(3 * ntri, dtype=np.uint32))
    pos = v["pos"][:, :3]
    tri = pos.reshape(-1, 3, 3).astype(np.float64)
    # no zero-area triangle survives the device-side filter
    assert np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1).min() > 1e-12
    # every vertex lies on the iso-surface (linear interpolation along cell edges of a distance field with sharp
    # features: within a fraction of a cell); normals point out of the solid and have unit length
    assert np.abs(reference_density(pos.astype(np.float64))).max() < 0.2 * cell
    n = v["normal"][:, :3].astype(np.float64)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-4)
    # a closed, consistently oriented surface: shared edge vertices are bit-identical between cells (canonical
    # interpolation order), so directed edges pair up exactly; the few that do not sit next to dropped slivers
    ids = np.unique(pos.view(np.uint32).reshape(-1, 3), axis=0, return_inverse=True)[1].reshape(-1, 3)
    e = np.concatenate([ids[:, [0, 1]], ids[:, [1, 2]], ids[:, [2, 0]]]).astype(np.int64)
    fwd, rev = e[:, 0] * (1 << 32) + e[:, 1], e[:, 1] * (1 << 32) + e[:, 0]
    matched = np.isin(fwd, rev).mean()
    assert matched > 0.995, matched
    # the same surface as the independent host generator (marching tetrahedra, float64 numpy), at a third to a half of the triangles
    ref = extract_isosurface(reference_density, lo, hi, res)
    ref = ref[np.linalg.norm(np.cross(ref[:, 1] - ref[:, 0], ref[:, 2] - ref[:, 0]), axis=1) > 1e-7]
    assert 0.2 < ntri / len(ref) < 0.65, (ntri, len(ref))

    from scipy.spatial import cKDTree

    a, b = tri.reshape(-1, 3), ref.reshape(-1, 3)
    assert cKDTree(b).query(a)[0].max() < 1.0 * cell and cKDTree(a).query(b)[0].max() < 1.0 * cell
    # the extracted mesh renders like any other: feed the very same triangles to the oracle
    from rust_renderer_amd.camera import Camera
    cpu = oa.OracleRenderer(W, H, threads=3)
    cpu.add_mesh(v, idx, rr.make_material(base_color=(0.8, 0.8, 0.8, 1.0), diffuse_map=cpu.default_diffuse_map()), None)
    cam = Camera((27.0, 19.0, 33.0), (16.0, 14.0, 16.0), 60.0, W / H, 0.01, 1000.0)
    for r in (gpu, cpu):
        r.initialize_raytracing()
        loop = rr.FrameLoop(r, rr.default_view(cam, W, H))
        loop.view.lights_enabled = 0
        for _ in range(2):
            loop.frame(rr.PASS_REFERENCE_PT)
    assert per_pixel_l2(gpu.read_accumulation(), cpu.read_accumulation()) <= L2_TOL
    assert list(gpu.get_stats().rays) == list(cpu.get_stats().rays)
    # deterministic output order (device scan, not an atomic append): a second extraction gives the same bytes
    again = rr.Renderer(8, 8)
    again.set_option("iso_reference_triangulation", 0)
    m2, n2 = again.add_isosurface_mesh(res, lo, hi)
    assert n2 == ntri and again.read_mesh(m2)[0].tobytes() == v.tobytes()
    # the default extraction (the reference's triangles, zero-area ones included) renders against the oracle on the same triangles too
    ref_gpu = rr.Renderer(W, H)
    rmesh, rtri = ref_gpu.add_isosurface_mesh(res, lo, hi)
    assert rtri >= ntri
    rv, ridx = ref_gpu.read_mesh(rmesh)
    ref_cpu = oa.OracleRenderer(W, H, threads=3)
    ref_cpu.add_mesh(rv, ridx, rr.make_material(base_color=(0.8, 0.8, 0.8, 1.0), diffuse_map=ref_cpu.default_diffuse_map()), None)
    for r in (ref_gpu, ref_cpu):
        r.initialize_raytracing()
        loop = rr.FrameLoop(r, rr.default_view(cam, W, H))
        loop.view.lights_enabled = 0
        for _ in range(2):
            loop.frame(rr.PASS_REFERENCE_PT)
    assert per_pixel_l2(ref_gpu.read_accumulation(), ref_cpu.read_accumulation()) <= L2_TOL
    assert list(ref_gpu.get_stats().rays) == list(ref_cpu.get_stats().rays)
    # the animated sphere (marching_cubes.comp:90) adds surface; nothing crossing the iso value adds no mesh
    _, with_sphere = rr.Renderer(8, 8).add_isosurface_mesh(res, lo, hi, time=3.0)
    assert with_sphere > ntri
    none, zero = rr.Renderer(8, 8).add_isosurface_mesh(4, 100.0, 101.0)
    assert none is None and zero == 0


def test_cook_torrance_extension_matches_oracle(atrium):
    """material type 4: same arithmetic on both sides, bit for bit once the sky is off"""
    scene = rr.scenes.sponza_class_scene(detail=0.1, tex_size=16, with_spheres=True, num_lights=8, sphere_subdivisions=2, cook_torrance=True)
    assert any(m.material_type == rr.types.PBR for model, _ in scene.models for m in model.meshes)
    W, H = 80, 48
    gpu, cpu = make_pair(scene, W, H, threads=3)
    for r in (gpu, cpu):
        run_frames(r, scene, W, H, 3, rr.PASS_ALL, sky_enabled=0)
    assert np.array_equal(gpu.read_accumulation().view(np.uint32), cpu.read_accumulation().view(np.uint32))
    assert list(gpu.get_stats().rays) == list(cpu.get_stats().rays)
    assert gpu.read_accumulation()[..., :3].max() > 0.0


@pytest.mark.parametrize("kind", [1, 2])
def test_device_build_falls_back_to_the_host_builder_on_overflowing_geometry(kind):
    """finite coordinates around 1e19: surface-area products overflow to inf, a PLOC round finds no pair to merge - the device
    build hands such a scene to the host builder instead of failing (the host tree handles it); hits equal the host build's"""
    from rust_renderer_amd.camera import Camera
    from rust_renderer_amd.scenes import Mesh, Model, Scene, _pack_vertices
    rng = np.random.default_rng(11)
    c = rng.uniform(-1, 1, (400, 1, 3)) * 3e19
    pos = (c + rng.uniform(-1, 1, (400, 3, 3)) * 2e18).reshape(-1, 3).astype(np.float32)
    m = Mesh(_pack_vertices(pos, np.tile(np.float32([0, 0, 1]), (len(pos), 1)), np.zeros((len(pos), 2), np.float32)), np.arange(len(pos), dtype=np.uint32))
    scene = Scene("huge", [(Model([m], []), None)], [], Camera((0, 0, 4), (0, 0, 0), 60.0, 1.0, 0.01, 100.0))
    rays = np.empty((4000, 8), dtype=np.float32)
    rays[:, 0:3] = rng.uniform(-1, 1, (4000, 3)) * 3e19
    rays[:, 4:7] = pos[rng.integers(0, len(pos), 4000)] - rays[:, 0:3]
    rays[:, 3], rays[:, 7] = 1e-3, 10.0
    host = scene.upload(rr.Renderer(8, 8))
    dev = rr.Renderer(8, 8)
    dev.set_option("device_build", kind)
    scene.upload(dev)
    assert dev.get_stats().bvh_triangles == 400  # built, not refused
    for a, b in zip(host.trace_closest(rays), dev.trace_closest(rays)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))  # (at this scale the triangle test itself overflows: all miss, on both)
