"""One frame per call (renderers/mod.rs:357, main.rs:460-471): bounces 1 .. of a lone frame inside one persistent kernel
(csrc/kernels.hip k_path_fused: a wavefront per block, the paths' state in place) must give the words the wavefront of launches
gives - radiance, output, ray counts - on every kind of frame the library renders, and the oracle's image."""
import numpy as np
import pytest

import rust_renderer_amd as rr
from util import L2_TOL, make_pair, per_pixel_l2, run_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cornell():
    return rr.scenes.cornell_scene(subdivisions=2, tex_size=16)


@pytest.fixture(scope="module")
def atrium():
    return rr.scenes.sponza_class_scene(detail=0.12, tex_size=32, with_spheres=True, num_lights=64, sphere_subdivisions=2)


@pytest.fixture(scope="module")
def rtiow():
    return rr.scenes.rtiow_scene(subdivisions=3)


def pair(scene, W, H, options=(), tile=None):
    """the same scene twice: the fused kernel on every frame (-1: also with frames in flight) / the wavefront of launches"""
    out = []
    for fused in (-1, 0):
        r = scene.upload(rr.Renderer(W, H))
        r.set_option("fused_bounces", fused)
        for k, v in options:
            r.set_option(k, v)
        if tile:
            r.set_tile_partition(*tile)
        out.append(r)
    return out


def same_frames(a, b):
    assert np.array_equal(a.read_accumulation().view(np.uint32), b.read_accumulation().view(np.uint32))
    assert np.array_equal(a.read_output_bgra8(), b.read_output_bgra8())
    sa, sb = a.get_stats(), b.get_stats()
    assert list(sa.rays) == list(sb.rays)
    assert (sa.closest_hits, sa.misses, sa.sun_tree_rays) == (sb.closest_hits, sb.misses, sb.sun_tree_rays)


@pytest.mark.parametrize("bounces,spp", [(2, 1), (3, 2), (5, 1), (9, 1)])
def test_fused_equals_wavefront_sun(atrium, bounces, spp):
    """sky + sun shadow rays (the bench's frame), with the sun grid and - bounce 0's sun rays inside the kernel - without"""
    W, H = 192, 108
    for opts in ((), (("sun_grid", 0),), (("camera_grid", 0), ("sun_grid_inline_max_mb", 4096))):
        fused, wave = pair(atrium, W, H, opts)
        for r in (fused, wave):
            run_frames(r, atrium, W, H, 3, rr.PASS_REFERENCE_PT, num_bounces=bounces, samples_per_frame=spp)
        same_frames(fused, wave)


def test_fused_equals_wavefront_lights_and_reservoirs(atrium):
    """point lights through the reservoirs: a light ray per scattered path behind its sun ray, in the reference's order"""
    W, H = 160, 90
    for split in (0, 1):
        fused, wave = pair(atrium, W, H, (("full_frame_restir", split),))
        for r in (fused, wave):
            run_frames(r, atrium, W, H, 4, rr.PASS_ALL, use_ris_light_sampling=1)
        same_frames(fused, wave)
        for which in range(3):
            assert np.array_equal(fused.read_reservoirs(which), wave.read_reservoirs(which))


def test_fused_equals_wavefront_lights_no_sun(cornell):
    W, H = 96, 80
    fused, wave = pair(cornell, W, H)
    for r in (fused, wave):
        run_frames(r, cornell, W, H, 3, rr.PASS_REFERENCE_PT, sun_shadow_enabled=0)
    same_frames(fused, wave)


def test_fused_equals_wavefront_materials_and_counters(rtiow):
    """metal, glass and a light source (paths that end at a hit), the ground plane (no sun grid: every sun ray is the tree's), and the
    visit counters on"""
    W, H = 128, 128
    fused, wave = pair(rtiow, W, H, (("count_visits", 1),))
    for r in (fused, wave):
        run_frames(r, rtiow, W, H, 3, rr.PASS_REFERENCE_PT)
    same_frames(fused, wave)
    sf, sw = fused.get_stats(), wave.get_stats()
    assert sf.nodes_visited > 0 and sf.shadow_nodes_visited > 0
    # the bounce rays' walks are the same walks; the shadow rays' differ in order only (closest-hit order with an early exit
    # against the visibility walk's), never in their answers
    assert sf.tris_tested == pytest.approx(sw.tris_tested, rel=0.05) and sf.nodes_visited == pytest.approx(sw.nodes_visited, rel=0.05)


def test_fused_with_tile_partition(atrium):
    W, H = 192, 108
    fused, wave = pair(atrium, W, H, tile=(1, 3, 16))
    for r in (fused, wave):
        run_frames(r, atrium, W, H, 3, rr.PASS_REFERENCE_PT)
    same_frames(fused, wave)


def test_fused_against_the_oracle(atrium):
    W, H = 128, 72
    gpu, cpu = make_pair(atrium, W, H)
    gpu.set_option("fused_bounces", -1)
    for r in (gpu, cpu):
        run_frames(r, atrium, W, H, 2, rr.PASS_REFERENCE_PT)
    assert per_pixel_l2(gpu.read_accumulation(), cpu.read_accumulation()) <= L2_TOL
    assert list(gpu.get_stats().rays) == list(cpu.get_stats().rays)


def test_fused_only_while_no_frame_is_in_flight(atrium):
    """the default (1): a caller that waits for its frames gets the fused kernel (one traversal launch per frame), a caller that
    keeps frames in flight the wavefront (one per bounce) - the same image either way"""
    W, H = 640, 360
    r = atrium.upload(rr.Renderer(W, H))
    r.set_option("time_kernels", 1)
    r.set_option("camera_grid", 0)
    loop = rr.FrameLoop(r, atrium.make_view(W, H))
    loop.frame(rr.PASS_REFERENCE_PT)
    r.synchronize()
    r.reset_stats()
    for _ in range(4):
        loop.frame(rr.PASS_REFERENCE_PT)
        r.synchronize()
    waited = r.get_stats().trace_closest_launches
    assert waited == 4 * 2  # bounce 0's launch + the fused kernel
    r.set_option("fused_bounces", 0)
    r.reset_stats()
    for _ in range(4):
        loop.frame(rr.PASS_REFERENCE_PT)
        r.synchronize()
    assert r.get_stats().trace_closest_launches == 4 * 5
