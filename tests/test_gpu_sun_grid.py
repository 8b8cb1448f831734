"""-m gpu: the sun-direction visibility grid (csrc/sun_grid.cpp + k_trace_sun_grid) against the any-hit tree walk it replaces
for the sun shadow rays of reference.rgen:63-79 and against the oracle: same images bit for bit, same ray counts, for every
octant and for axis-aligned suns (walls edge-on), across direction changes, refits and device-built trees."""
import numpy as np
import pytest

import oracle_api as oa
import rust_renderer_amd as rr
from util import L2_TOL, make_pair, per_pixel_l2, run_frames, torture_scene

pytestmark = pytest.mark.gpu

SUNS = [(0.0, 0.9, 0.15), (0.3, 0.8, 0.2), (-0.3, -0.8, 0.2), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.7, 0.7, 0.0), (1e-4, 1.0, 0.0), (-0.5, 0.1, 0.85)]


@pytest.fixture(scope="module")
def atrium():
    return rr.scenes.sponza_class_scene(detail=0.12, tex_size=32, with_spheres=True, num_lights=0, sphere_subdivisions=2)


@pytest.fixture(scope="module")
def cornell():
    return rr.scenes.cornell_scene(subdivisions=2, tex_size=16)


def render(renderer, scene, W, H, sun, frames=2, **flags):
    loop = rr.FrameLoop(renderer, scene.make_view(W, H, sun_shadow_enabled=1, **flags))
    loop.view.sun_dir[:] = list(sun)
    for _ in range(frames):
        loop.frame(rr.PASS_REFERENCE_PT)
    return loop


@pytest.mark.parametrize("sun", SUNS)
def test_grid_equals_tree_walk_and_oracle(atrium, sun):
    W, H = 128, 72
    grid = atrium.upload(rr.Renderer(W, H))
    tree = atrium.upload(rr.Renderer(W, H))
    tree.set_option("sun_grid", 0)
    cpu = atrium.upload(oa.OracleRenderer(W, H))
    for r in (grid, tree, cpu):
        render(r, atrium, W, H, sun, frames=2, sky_enabled=0, lights_enabled=0)
    g, t = grid.get_stats(), tree.get_stats()
    axis_aligned = sorted(abs(x) for x in sun)[1] < 1e-3  # walls edge-on to the sun: the builder may refuse (too much surface in long-list cells)
    assert t.sun_grid_cells == 0 and (axis_aligned or (g.sun_grid_cells > 0 and g.sun_grid_entries > 0)), "the first renderer really went through the grid"
    a = grid.read_accumulation()
    assert np.array_equal(a.view(np.uint32), tree.read_accumulation().view(np.uint32))
    assert np.array_equal(a.view(np.uint32), cpu.read_accumulation().view(np.uint32))
    assert list(g.rays) == list(t.rays) and list(g.rays)[:4] == list(cpu.get_stats().rays)[:4]
    assert g.rays[rr.RAY_SUN_SHADOW] > 0


def test_grid_on_all_scene_kinds_with_sky(atrium, cornell):
    """the whole path (sky on, every material type, lights) with the grid against the oracle, and against the tree walk bit for bit"""
    for scene, (W, H) in ((cornell, (96, 64)), (torture_scene(), (24, 24)), (rr.scenes.rtiow_scene(2), (64, 64))):
        grid, cpu = make_pair(scene, W, H)
        tree = scene.upload(rr.Renderer(W, H))
        tree.set_option("sun_grid", 0)
        for r in (grid, tree, cpu):
            run_frames(r, scene, W, H, 3, rr.PASS_ALL)
        assert np.array_equal(grid.read_accumulation().view(np.uint32), tree.read_accumulation().view(np.uint32)), scene.name
        assert per_pixel_l2(grid.read_accumulation() / 3, cpu.read_accumulation() / 3) <= L2_TOL, scene.name
        assert list(grid.get_stats().rays) == list(tree.get_stats().rays) == list(cpu.get_stats().rays), scene.name


def test_counted_visits_drop(atrium):
    """what the grid is for: a cell look-up and a few triangle tests per sun ray instead of a tree walk"""
    W, H = 160, 90
    out = {}
    for name, on in (("grid", 1), ("tree", 0)):
        r = atrium.upload(rr.Renderer(W, H))
        r.set_option("sun_grid", on)
        r.set_option("count_visits", 1)
        render(r, atrium, W, H, SUNS[0], frames=2, sky_enabled=1, lights_enabled=0)
        s = r.get_stats()
        out[name] = (s.shadow_nodes_visited / s.rays[rr.RAY_SUN_SHADOW], s.shadow_tris_tested / s.rays[rr.RAY_SUN_SHADOW])
    assert out["grid"][0] == 1.0, "one cell per ray"
    assert out["grid"][1] < 6.0 and out["tree"][0] > 5.0, out


def test_direction_changes_and_settling(atrium):
    """frame 1 builds the grid for direction A; a frame with direction B walks the tree (a sun that moves every frame must not
    rebuild every frame); the second frame in a row with B rebuilds. Every frame equals the oracle's."""
    W, H = 96, 54
    gpu, cpu = make_pair(atrium, W, H)
    A, B, C = (0.0, 0.9, 0.15), (0.4, 0.7, -0.3), (-0.2, 0.5, 0.6)
    loops = [rr.FrameLoop(r, atrium.make_view(W, H, sun_shadow_enabled=1, sky_enabled=0, lights_enabled=0)) for r in (gpu, cpu)]
    builds = []
    for sun in (A, A, B, C, B, B, B, A, A):
        for loop in loops:
            loop.view.sun_dir[:] = list(sun)
            loop.frame(rr.PASS_REFERENCE_PT)
            loop.reset()  # every frame stands alone
        s = gpu.get_stats()
        builds.append(round(s.sun_grid_build_ms, 4))
        assert np.array_equal(gpu.read_accumulation().view(np.uint32), cpu.read_accumulation().view(np.uint32)), sun
    # builds happened at frames 0 (A), 5 (B asked twice in a row) and 8 (A again, twice in a row): the timer changes exactly there
    changed = [i for i in range(1, len(builds)) if builds[i] != builds[i - 1]]
    assert changed == [5, 8], (changed, builds)


def test_refit_and_device_build_invalidate_the_grid(cornell):
    W, H = 80, 60
    gpu, cpu = make_pair(cornell, W, H)
    for r in (gpu, cpu):
        run_frames(r, cornell, W, H, 1, rr.PASS_REFERENCE_PT)
    first = gpu.get_stats().sun_grid_entries
    move = rr.transform3x4((0.3,) * 3, (0.1, 0.9, 0.1))
    n = cornell.num_meshes
    for r in (gpu, cpu):
        r.set_instance_transform(n - 2, move)  # the metal sphere now hangs in the air and casts another shadow
        r.rebuild_tlas()
        r.reset_accumulation()
        run_frames(r, cornell, W, H, 2, rr.PASS_REFERENCE_PT)
    assert per_pixel_l2(gpu.read_accumulation(), cpu.read_accumulation()) <= L2_TOL
    assert list(gpu.get_stats().rays) == list(cpu.get_stats().rays)
    assert gpu.get_stats().sun_grid_entries > 0 and first > 0
    for kind in (1, 2):
        dev = cornell.upload(rr.Renderer(W, H))
        dev.set_option("device_build", kind)
        dev.initialize_raytracing()
        ref = cornell.upload(rr.Renderer(W, H))
        for r in (dev, ref):
            run_frames(r, cornell, W, H, 2, rr.PASS_REFERENCE_PT)
        assert dev.get_stats().sun_grid_cells > 0
        assert np.array_equal(dev.read_accumulation().view(np.uint32), ref.read_accumulation().view(np.uint32)), kind


def test_instances_moving_every_frame_never_rebuild_the_grid(cornell):
    """rebuild_tlas = 1 with an instance that moves every frame (main.rs:392,526): every frame refits, none builds a grid
    (a build costs a thousand frames' saving); the first frame at rest brings the grid back. Every frame equals the oracle's."""
    W, H = 80, 60
    gpu, cpu = make_pair(cornell, W, H)
    loops = [rr.FrameLoop(r, cornell.make_view(W, H, sun_shadow_enabled=1, lights_enabled=0)) for r in (gpu, cpu)]
    n = cornell.num_meshes
    builds = []
    for k in range(8):
        moved = k in (2, 3, 4, 5)
        for r, loop in zip((gpu, cpu), loops):
            if moved:
                r.set_instance_transform(n - 2, rr.transform3x4((0.3,) * 3, (0.1 + 0.05 * k, 0.9, 0.1)))
            loop.view.rebuild_tlas = 1 if moved else 0
            loop.frame(rr.PASS_REFERENCE_PT)
            loop.reset()
        builds.append(round(gpu.get_stats().sun_grid_build_ms, 4))
        assert per_pixel_l2(gpu.read_accumulation(), cpu.read_accumulation()) <= L2_TOL, k
        assert list(gpu.get_stats().rays) == list(cpu.get_stats().rays), k
    changed = [i for i in range(1, len(builds)) if builds[i] != builds[i - 1]]
    assert changed == [6], (changed, builds)  # frames 2-5 move; frame 6 is the second in a row with frame 5's geometry: the rebuild


def test_rays_beyond_the_dense_extent_walk_the_tree(atrium):
    """a long strip of ground that leaves the atrium through its end wall (two triangles, little area: the grid is not refused as
    a whole): rays that start out there land in border cells and are handed to the tree walk (queue 3); the image equals the
    tree-only render and the oracle's"""
    import copy

    from rust_renderer_amd.scenes import Mesh, Model
    W, H = 96, 64
    scene = copy.copy(atrium)
    strip = Mesh(*rr.scenes.quad((16.5, 0.02, -1.0), (0, 0, 2.0), (40.0, 0, 0), 1, 1), base_color=(0.5, 0.5, 0.5, 1.0), name="strip")
    scene.models = list(atrium.models) + [(Model([strip], []), None)]
    scene.camera = rr.camera.Camera((60.0, 6.0, 3.0), (35.0, 0.0, 0.0), 60.0, W / H, 0.01, 1000.0)
    grid, cpu = make_pair(scene, W, H)
    tree = scene.upload(rr.Renderer(W, H))
    tree.set_option("sun_grid", 0)
    for r in (grid, tree, cpu):
        render(r, scene, W, H, (0.3, 0.8, 0.2), frames=2, sky_enabled=0, lights_enabled=0)
    assert grid.get_stats().sun_grid_cells > 0
    a = grid.read_accumulation()
    assert (a[..., :3] > 0).mean() > 0.05, "the strip outside the building is in the picture and lit"
    assert np.array_equal(a.view(np.uint32), tree.read_accumulation().view(np.uint32))
    assert np.array_equal(a.view(np.uint32), cpu.read_accumulation().view(np.uint32))
    assert list(grid.get_stats().rays) == list(tree.get_stats().rays)


def test_budget_refusal_falls_back_to_the_tree(atrium):
    W, H = 64, 36
    gpu, cpu = make_pair(atrium, W, H)
    gpu.set_option("sun_grid_max_mb", 1)  # 131,072 entries: far too few for this scene at any useful cell size
    for r in (gpu, cpu):
        render(r, atrium, W, H, SUNS[0], frames=1, sky_enabled=0, lights_enabled=0)
    assert np.array_equal(gpu.read_accumulation().view(np.uint32), cpu.read_accumulation().view(np.uint32))


def test_tile_partition_and_batches_with_the_grid(atrium):
    W, H = 192, 108
    ref = atrium.upload(rr.Renderer(W, H))
    ref.set_option("sun_grid", 0)
    rr.FrameLoop(ref, atrium.make_view(W, H)).frames(6, rr.PASS_REFERENCE_PT)
    whole = atrium.upload(rr.Renderer(W, H))
    whole.set_option("batch_frames", 4)
    rr.FrameLoop(whole, atrium.make_view(W, H)).frames(6, rr.PASS_REFERENCE_PT)
    assert np.array_equal(ref.read_accumulation().view(np.uint32), whole.read_accumulation().view(np.uint32))
    group = atrium.upload(rr.MultiGpuRenderer(W, H, devices=[0, 0, 0], tile_size=32))
    rr.FrameLoop(group, atrium.make_view(W, H)).frames(6, rr.PASS_REFERENCE_PT)
    assert np.array_equal(ref.read_accumulation().view(np.uint32), group.read_accumulation().view(np.uint32))
    assert list(ref.get_stats().rays) == list(group.get_stats().rays)


def test_scene_deeper_than_tmax_along_the_sun():
    """the cell's cover depth must respect the rays' tmax = 10000 (rgen:45,66-67). A floor at y = 0 under a roof at
    y = 20000 with the sun straight up: the roof covers every cell, but a ray from the floor would meet it at t = 19999 > tmax -
    the tree walk and the oracle leave the floor lit, and so must the grid. A slab at y = 8000 does shade its part of the floor."""
    from rust_renderer_amd.scenes import Mesh, Model, Scene, quad
    from rust_renderer_amd.camera import Camera

    meshes = [Mesh(*quad((-4, 0, -4), (0, 0, 8), (8, 0, 0), 8, 8), base_color=(0.8, 0.8, 0.8, 1.0), name="floor"),
              Mesh(*quad((-40, 20000, -40), (80, 0, 0), (0, 0, 80), 1, 1), base_color=(0.5, 0.5, 0.5, 1.0), name="roof"),
              Mesh(*quad((-1, 8000, -1), (2, 0, 0), (0, 0, 2), 1, 1), base_color=(0.5, 0.5, 0.5, 1.0), name="slab")]
    cam = Camera((0.0, 6.0, 7.0), (0.0, 0.0, 0.0), 60.0, 1.0, 0.01, 1000.0)
    scene = Scene("deep", [(Model(meshes, []), None)], [], cam, dict(sky_enabled=0, sun_shadow_enabled=1, lights_enabled=0, num_bounces=2))
    W, H = 96, 96
    grid = scene.upload(rr.Renderer(W, H))
    grid.set_option("sun_grid_force", 1)  # margins at |y| = 20000 are half a unit wide: long lists, and this test wants the grid
    tree = scene.upload(rr.Renderer(W, H))
    tree.set_option("sun_grid", 0)
    cpu = scene.upload(oa.OracleRenderer(W, H))
    for r in (grid, tree, cpu):
        render(r, scene, W, H, (0.0, 1.0, 0.0), frames=2)
    assert grid.get_stats().sun_grid_cells > 0, "the grid was refused: the test does not test it"
    a = grid.read_accumulation()
    assert np.array_equal(a.view(np.uint32), tree.read_accumulation().view(np.uint32))
    assert np.array_equal(a.view(np.uint32), cpu.read_accumulation().view(np.uint32))
    lit = a[..., 0] > 0
    assert lit.any() and not lit.all()  # the floor is lit except under the slab


@pytest.mark.parametrize("sun", [SUNS[0], SUNS[2], SUNS[8], SUNS[3]])
def test_device_built_grid_equals_the_host_built_one(atrium, sun):
    """round 4: the grid is built on the device (sun_grid_build.hip) by default. The host builder stays the reference
    implementation (its margins are held against brute force on the CPU); run on the same packets and the same raster, the two
    must bin every packet into the same cells, find the same cover depth in every cell, bit for bit, and sort the lists a ray may
    walk into the same order."""
    W, H = 96, 54
    r = atrium.upload(rr.Renderer(W, H))
    r.set_option("sun_grid_force", 1)
    render(r, atrium, W, H, sun, frames=1, sky_enabled=0, lights_enabled=0)
    s = r.get_stats()
    assert s.sun_grid_cells > 0 and s.sun_grid_entries > 0, "no grid was built"
    d = r.sun_grid_compare_builders()
    assert d["entries_device"] == d["entries_host"] == s.sun_grid_entries
    assert d["cells_length_differs"] == 0 and d["cells_list_differs"] == 0 and d["cells_cover_differs"] == 0, d
    assert d["walkable_cells"] > 1000
    assert s.sun_grid_build_ms < d["host_build_us"] / 1000.0, (s.sun_grid_build_ms, d["host_build_us"])


def test_host_builder_is_still_selectable(atrium):
    W, H = 96, 54
    dev, host, tree = (atrium.upload(rr.Renderer(W, H)) for _ in range(3))
    host.set_option("sun_grid_build", 0)
    tree.set_option("sun_grid", 0)
    for r in (dev, host, tree):
        render(r, atrium, W, H, SUNS[1], frames=2, sky_enabled=1, lights_enabled=0)
    assert dev.get_stats().sun_grid_cells > 0 and host.get_stats().sun_grid_cells > 0
    a = dev.read_accumulation().view(np.uint32)
    assert np.array_equal(a, host.read_accumulation().view(np.uint32)) and np.array_equal(a, tree.read_accumulation().view(np.uint32))
    assert list(dev.get_stats().rays) == list(host.get_stats().rays) == list(tree.get_stats().rays)


@pytest.mark.parametrize("options", [{"sun_grid_inline_max_mb": 0}, {"sun_grid_inline_max_mb": 8192}, {"sun_grid_coarse": 0}, {"sun_grid_coarse": 4, "sun_grid_inline_max_mb": 8192},
                                     {"sun_grid_inline_max_mb": 1}, {"sun_grid_density": 24, "sun_grid_max_walk": 8, "sun_grid_inline_max_mb": 8192}])
@pytest.mark.parametrize("sun", [SUNS[0], SUNS[2], SUNS[8]])
def test_variants_of_the_grid_walk_change_nothing(atrium, sun, options):
    """the two forms of the lists - plain (8-byte entries beside the packet array: the default while the 64-byte records would exceed
    four times the packet array, as on this scene) and as 64-byte records that carry their packet (sun_grid_inline_max_mb raised) -,
    with and without the coarse cover: same images, ray counts and counted visits as the default path and as the tree walk"""
    W, H = 160, 90
    out = []
    for opts in ({}, options, {"sun_grid": 0}):
        r = atrium.upload(rr.Renderer(W, H))
        r.set_option("count_visits", 1)
        for k, v in opts.items():
            r.set_option(k, v)
        render(r, atrium, W, H, sun, frames=3, sky_enabled=1, lights_enabled=0)
        out.append((r.read_accumulation().view(np.uint32), r.get_stats()))
    (a0, s0), (a1, s1), (a2, s2) = out
    assert s0.sun_grid_cells > 0 and s1.sun_grid_cells > 0 and s2.sun_grid_cells == 0
    assert np.array_equal(a0, a1) and np.array_equal(a0, a2)
    assert list(s0.rays) == list(s1.rays) == list(s2.rays)
    if "sun_grid_density" not in options:  # (another raster: other lists)
        # the same cells looked up, the same packets tested, the same rays handed to the tree, the same rays answered by a cover depth
        assert (s0.shadow_nodes_visited, s0.shadow_tris_tested, s0.sun_tree_rays, s0.sun_covered_rays) == (s1.shadow_nodes_visited, s1.shadow_tris_tested, s1.sun_tree_rays, s1.sun_covered_rays)


def test_variants_with_lights_and_batches(atrium):
    """config-2 style frames (lights on, reservoir passes, batches of frames): the lists as records that carry their packet against the plain lists"""
    scene = rr.scenes.sponza_class_scene(detail=0.12, tex_size=32, with_spheres=True, num_lights=48, sphere_subdivisions=2)
    W, H = 128, 72
    images = []
    for opts in ({}, {"sun_grid_inline_max_mb": 8192}):
        r = scene.upload(rr.Renderer(W, H))
        for k, v in opts.items():
            r.set_option(k, v)
        loop = rr.FrameLoop(r, scene.make_view(W, H, sun_shadow_enabled=1, sky_enabled=1, lights_enabled=1))
        loop.frame(rr.PASS_ALL)
        loop.frame(rr.PASS_ALL)
        loop.frames(6, rr.PASS_ALL)
        images.append((r.read_accumulation().view(np.uint32), list(r.get_stats().rays), r.get_stats().sun_grid_cells))
    assert images[0][2] > 0 and images[1][2] > 0
    assert np.array_equal(images[0][0], images[1][0]) and images[0][1] == images[1][1]
