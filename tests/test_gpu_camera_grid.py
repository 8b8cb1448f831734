"""-m gpu: the per-camera grid (csrc/sun_grid_build.hip k_pg_project + kernels.hip k_trace_camera_grid / k_gbuffer_camera_grid)
against the tree walk it replaces for the primary rays of reference.rgen:31-47 and the G-buffer cast (gbuffer.rs:11-52): the
same hit for every ray - accumulation images, G-buffer positions and reservoirs bit for bit, ray counts equal - for cameras
inside and outside the geometry, touching it, lying in the plane of triangles, with partial frames, partitions and batches."""
import numpy as np
import pytest

import oracle_api as oa
import rust_renderer_amd as rr
from rust_renderer_amd.camera import Camera
from util import torture_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def atrium():
    return rr.scenes.sponza_class_scene(detail=0.12, tex_size=32, with_spheres=True, num_lights=4, sphere_subdivisions=2)


@pytest.fixture(scope="module")
def cornell():
    return rr.scenes.cornell_scene(subdivisions=2, tex_size=16)


def pair(scene, W, H):
    grid, tree = scene.upload(rr.Renderer(W, H)), scene.upload(rr.Renderer(W, H))
    tree.set_option("camera_grid", 0)
    return grid, tree


def same(grid, tree, what=""):
    a, b = grid.read_accumulation(), tree.read_accumulation()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"accumulation differs {what}: {int((a != b).any(axis=-1).sum())} pixels"
    assert np.array_equal(grid.read_gbuffer_position().view(np.uint32), tree.read_gbuffer_position().view(np.uint32)), f"G-buffer differs {what}"
    for k in range(3):
        assert np.array_equal(grid.read_reservoirs(k), tree.read_reservoirs(k)), f"reservoirs {k} differ {what}"
    assert list(grid.get_stats().rays) == list(tree.get_stats().rays), what


CAMERAS = [
    ((-10.28, 2.10, -0.18), (0.0, 0.5, 0.0)),      # the reference's Sponza camera (scenes.rs:107-110)
    ((0.0, 1.0, 0.0), (5.0, 1.2, 1.0)),            # mid-atrium, along the nave
    ((3.0, 6.5, 1.0), (0.0, 0.0, 0.0)),            # from above, looking down
    ((-2.0, 0.02, 0.5), (4.0, 0.4, 0.5)),          # two centimetres above the floor: grazing views of it
    ((-40.0, 12.0, 25.0), (0.0, 2.0, 0.0)),        # from outside: the whole scene in a corner of the frame
    ((0.0, 0.0, 0.0), (1.0, 0.0, 0.3)),            # IN the plane of the floor (y = 0): every floor triangle edge-on
]


@pytest.mark.parametrize("cam", range(len(CAMERAS)))
def test_grid_equals_tree_walk(atrium, cam):
    W, H = 160, 90
    grid, tree = pair(atrium, W, H)
    atrium.camera = Camera(CAMERAS[cam][0], CAMERAS[cam][1], 60.0, W / H, 0.01, 1000.0)
    for r in (grid, tree):
        loop = rr.FrameLoop(r, atrium.make_view(W, H))
        loop.frames(9, rr.PASS_ALL)  # the first frame goes alone (temporal pass), the batch of 8 builds the grid at once
        loop.frame(rr.PASS_ALL)
    g = grid.get_stats()
    assert tree.get_stats().camera_grid_cells == 0
    assert g.camera_grid_cells == W * H and g.camera_grid_entries > 0, "the first renderer did not go through the grid"
    same(grid, tree, f"camera {cam}")


def test_grid_equals_oracle(cornell):
    W, H = 96, 64
    gpu, cpu = cornell.upload(rr.Renderer(W, H)), cornell.upload(oa.OracleRenderer(W, H))
    for r in (gpu, cpu):
        loop = rr.FrameLoop(r, cornell.make_view(W, H, sky_enabled=0))
        for _ in range(4):
            loop.frame(rr.PASS_ALL)
    assert gpu.get_stats().camera_grid_cells == W * H
    assert np.array_equal(gpu.read_accumulation().view(np.uint32), cpu.read_accumulation().view(np.uint32))
    assert np.array_equal(gpu.read_gbuffer_position().view(np.uint32), cpu.read_gbuffer_position().view(np.uint32))
    assert list(gpu.get_stats().rays) == list(cpu.get_stats().rays)


def test_settling_moving_camera_and_geometry_changes(atrium):
    """a camera that moves every frame never builds a grid; at rest the second frame builds it; another camera or moved geometry
    drops it - and every frame equals the tree walk"""
    from rust_renderer_amd.api import transform3x4

    W, H = 128, 72
    grid, tree = pair(atrium, W, H)
    loops = []
    for r in (grid, tree):
        atrium.camera = Camera(CAMERAS[1][0], CAMERAS[1][1], 60.0, W / H, 0.01, 1000.0)
        loops.append(rr.FrameLoop(r, atrium.make_view(W, H)))

    def frame(eye, target):
        cam = Camera(eye, target, 60.0, W / H, 0.01, 1000.0)
        for loop in loops:
            loop.view.view[:] = rr.camera.to_glam(cam.get_view()).tolist()
            loop.view.inverse_view[:] = rr.camera.to_glam(rr.camera.inverse(cam.get_view())).tolist()
            loop.view.eye_pos[:3] = list(eye)
            loop.frame(rr.PASS_ALL)
        return grid.get_stats().camera_grid_cells

    for k in range(4):
        assert frame((0.1 * k, 1.0, 0.0), (5.0, 1.2, 1.0)) == 0, "a camera on the move builds nothing"
    assert frame((0.3, 1.0, 0.0), (5.0, 1.2, 1.0)) == W * H, "the second frame from one place builds the grid"
    assert frame((0.3, 1.0, 0.0), (5.0, 1.2, 1.0)) == W * H
    same(grid, tree, "camera at rest")
    assert frame((0.4, 1.0, 0.0), (5.0, 1.2, 1.0)) == 0, "another camera: the grid is not used"
    assert frame((0.4, 1.0, 0.0), (5.0, 1.2, 1.0)) == W * H
    for r in (grid, tree):
        r.set_instance_transform(0, transform3x4((1.0, 1.0, 1.0), (0.0, 0.15, 0.0)))
        r.refit_acceleration()
    assert frame((0.4, 1.0, 0.0), (5.0, 1.2, 1.0)) == 0, "moved geometry: the grid is not used"
    assert frame((0.4, 1.0, 0.0), (5.0, 1.2, 1.0)) == W * H
    same(grid, tree, "after the refit")


def test_all_scene_kinds_and_odd_frames(cornell):
    for scene, (W, H) in ((cornell, (97, 61)), (torture_scene(), (33, 24)), (rr.scenes.rtiow_scene(2), (64, 64))):
        grid, tree = pair(scene, W, H)
        for r in (grid, tree):
            loop = rr.FrameLoop(r, scene.make_view(W, H))
            for _ in range(3):
                loop.frame(rr.PASS_ALL)
        assert grid.get_stats().camera_grid_cells == W * H, scene.name
        same(grid, tree, scene.name)


@pytest.mark.parametrize("walk_whole", [0, 3, 512])
def test_long_lists_go_to_the_tree_or_are_walked_whole(atrium, walk_whole):
    """max_walk = 1: nearly every pixel's list is too long for the sorted walk. With camera_grid_walk_whole = 0 those pixels hand
    their rays to the tree walk (the listed-positions form of k_trace_closest), with 512 the grid kernel walks the unsorted lists
    whole (and the launch of the tree walk behind it is not made), with 3 some of each; same image either way"""
    W, H = 128, 72
    grid, tree = pair(atrium, W, H)
    grid.set_option("camera_grid_max_walk", 1)
    grid.set_option("camera_grid_walk_whole", walk_whole)
    for r in (grid, tree):
        loop = rr.FrameLoop(r, atrium.make_view(W, H))
        loop.frames(10, rr.PASS_ALL)
        loop.frame(rr.PASS_ALL)
    g = grid.get_stats()
    assert g.camera_grid_cells == W * H
    if walk_whole == 0:
        assert g.camera_tree_rays > 0.5 * g.rays[rr.RAY_PRIMARY] * 0.5
    elif walk_whole == 512:
        assert g.camera_tree_rays == 0
    else:
        assert 0 < g.camera_tree_rays < g.rays[rr.RAY_PRIMARY]
    same(grid, tree, f"max_walk 1, walk_whole {walk_whole}")


@pytest.mark.parametrize("spp", [3, 1])
def test_tile_partition_and_samples_per_frame(atrium, spp):
    """a rank's tiles of a partition (path ids dense over its owned pixels), several samples per frame (the raygen RNG word carried
    from sample to sample) and one (primary_implicit: the primary rays' state computed from the path id through the owned-pixel list)"""
    W, H = 128, 96
    grid, tree = pair(atrium, W, H)
    for r in (grid, tree):
        r.set_tile_partition(1, 3, 32)
        loop = rr.FrameLoop(r, atrium.make_view(W, H, samples_per_frame=spp))
        loop.frames(9, rr.PASS_REFERENCE_PT)
        loop.frame(rr.PASS_REFERENCE_PT)
    assert grid.get_stats().camera_grid_cells == W * H
    a, b = grid.read_accumulation(), tree.read_accumulation()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert list(grid.get_stats().rays) == list(tree.get_stats().rays)


def test_refused_when_the_lists_are_too_long(atrium):
    """a camera far outside sees the whole scene in a few pixels: their lists hold thousands of packets - the grid is refused by
    its mean list and the frames keep the tree walk"""
    W, H = 64, 36
    grid, tree = pair(atrium, W, H)
    grid.set_option("camera_grid_max_mean_list_x10", 20)
    atrium.camera = Camera((-400.0, 120.0, 250.0), (0.0, 2.0, 0.0), 60.0, W / H, 0.01, 10000.0)
    for r in (grid, tree):
        loop = rr.FrameLoop(r, atrium.make_view(W, H))
        loop.frames(9, rr.PASS_ALL)
    assert grid.get_stats().camera_grid_cells == 0
    same(grid, tree, "refused grid")


@pytest.mark.parametrize("max_walk", [48, 2])
def test_primary_rays_without_stored_state(atrium, max_walk):
    """option primary_implicit (default 1: with the camera grid and one sample per frame the origin and throughput planes of bounce 0
    are neither written nor read - the camera position, 1.0 and the RNG word recomputed from the payload seed stand in) against the
    stored form and against the tree walk, also with most rays handed to the tree (their origins are written after all)"""
    W, H = 160, 90
    out = []
    atrium.camera = Camera(CAMERAS[1][0], CAMERAS[1][1], 60.0, W / H, 0.01, 1000.0)  # (the fixture is shared: the test before leaves a camera whose grid is refused)
    for opts in ({}, {"primary_implicit": 0}, {"camera_grid": 0}):
        r = atrium.upload(rr.Renderer(W, H))
        r.set_option("camera_grid_max_walk", max_walk)
        r.set_option("camera_grid_walk_whole", 0)  # (most rays handed to the tree at max_walk 2: their records are written after all)
        for k, v in opts.items():
            r.set_option(k, v)
        loop = rr.FrameLoop(r, atrium.make_view(W, H, sky_enabled=1, sun_shadow_enabled=1, lights_enabled=1))
        loop.frames(9, rr.PASS_ALL)
        loop.frame(rr.PASS_ALL)
        out.append((r.read_accumulation().view(np.uint32), list(r.get_stats().rays), r.get_stats().camera_grid_cells))
    assert out[0][2] == W * H and out[1][2] == W * H and out[2][2] == 0
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][0], out[2][0])
    assert out[0][1] == out[1][1] == out[2][1]
