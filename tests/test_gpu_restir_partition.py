"""-m gpu: the reservoir passes partitioned by bands of rows (uh_set_restir_partition; SURVEY.md 8e alternative:
temporal_reuse.rgen:90-99 and spatial_reuse.rgen:40-60 read across any pixel partition). Everything is held against the
single-context render and the oracle bit for bit: the group's peer-copy exchange on one GPU (N contexts on device 0), the
rows a context computes, the RCCL link with one rank, a rank's share without an exchange."""
import numpy as np
import pytest

import oracle_api as oa
import rust_renderer_amd as rr
from util import make_pair, run_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def atrium():
    return rr.scenes.sponza_class_scene(detail=0.12, tex_size=32, with_spheres=True, num_lights=48, sphere_subdivisions=2)


def same_reservoirs(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


@pytest.mark.parametrize("ngpu,H", [(3, 100), (4, 72), (2, 31)])
def test_group_with_band_partition_equals_single_context_and_oracle(atrium, ngpu, H):
    """H = 100 over 3: bands of 34, 34, 32; 72 over 4: bands of 18 (every band reaches past its neighbours' neighbours);
    31 over 2: the halo covers the whole frame"""
    W = 96
    one, cpu = make_pair(atrium, W, H)
    group = atrium.upload(rr.MultiGpuRenderer(W, H, devices=[0] * ngpu, tile_size=16))
    for r in (one, group):
        rr.FrameLoop(r, atrium.make_view(W, H)).frames(21, rr.PASS_ALL)  # the first frame alone, then batches of 16 + 4: the exchange runs inside batched chains
    run_frames(cpu, atrium, W, H, 21, rr.PASS_ALL)
    assert np.array_equal(group.read_accumulation().view(np.uint32), one.read_accumulation().view(np.uint32))
    for which in range(3):
        got = group.read_reservoirs(which)
        assert same_reservoirs(got, one.read_reservoirs(which)), which
        assert same_reservoirs(got, cpu.read_reservoirs(which)), which
    assert (group.read_reservoirs(2)["M"] > 1).any()
    assert list(group.get_stats().rays) == list(one.get_stats().rays) == list(cpu.get_stats().rays)


def test_group_every_gpu_holds_the_whole_spatial_buffer(atrium):
    W, H, n = 64, 70, 3
    group = atrium.upload(rr.MultiGpuRenderer(W, H, devices=[0] * n, tile_size=16))
    one = atrium.upload(rr.Renderer(W, H))
    for r in (one, group):
        loop = rr.FrameLoop(r, atrium.make_view(W, H))
        for _ in range(3):
            loop.frame(rr.PASS_ALL)  # frame by frame: the un-batched path through the same phases
    ref = one.read_reservoirs(2)
    lib = group._lib
    import ctypes as C

    lib.uh_read_reservoirs.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    for i in range(n):
        out = np.empty((H, W), dtype=ref.dtype)
        assert lib.uh_read_reservoirs(lib.uh_mgpu_context(group._ctx, i), 2, out.ctypes.data) == 0
        assert same_reservoirs(out, ref), i


def test_group_option_off_runs_full_frame_passes(atrium):
    W, H = 64, 48
    group = atrium.upload(rr.MultiGpuRenderer(W, H, devices=[0, 0], tile_size=16))
    group.set_option("restir_partition", 0)
    one = atrium.upload(rr.Renderer(W, H))
    for r in (one, group):
        rr.FrameLoop(r, atrium.make_view(W, H)).frames(4, rr.PASS_ALL)
    assert np.array_equal(group.read_accumulation().view(np.uint32), one.read_accumulation().view(np.uint32))
    for which in range(3):
        assert same_reservoirs(group.read_reservoirs(which), one.read_reservoirs(which))
    assert list(group.get_stats().rays) == list(one.get_stats().rays)  # the replicated G-buffer cast is reported once


def test_rows_equal_the_checkers(atrium):
    for H, world in ((50, 3), (72, 4), (31, 2), (7, 8), (200, 8), (100, 1)):
        gpu, cpu = rr.Renderer(8, H), oa.OracleRenderer(8, H)
        for rank in range(world):
            for r in (gpu, cpu):
                r.set_restir_partition(rank, world)
            a, b = gpu.restir_rows(), cpu.restir_rows()
            mine, theirs = [getattr(a, f[0]) for f in a._fields_], [getattr(b, f[0]) for f in b._fields_]
            if b.band_rows == 0:
                assert a.band_rows == 0
                continue
            assert mine == theirs, (H, world, rank, mine, theirs)
        gpu.close()


def test_one_rank_of_four_without_exchange_first_frame(atrium):
    """what bench.py --emulate-world N --config 2 times: one rank's rows, nobody to exchange with. On the first frame (empty
    history everywhere) the rank's band of all three buffers is the whole-frame result's."""
    W, H, world = 96, 120, 4
    full = atrium.upload(rr.Renderer(W, H))
    run_frames(full, atrium, W, H, 1, rr.PASS_RESTIR)
    for rank in range(world):
        part = atrium.upload(rr.Renderer(W, H))
        part.set_restir_partition(rank, world)
        run_frames(part, atrium, W, H, 1, rr.PASS_RESTIR)
        rows = part.restir_rows()
        band = slice(rows.band_row0, rows.band_row0 + rows.band_rows)
        for which in range(3):
            assert same_reservoirs(part.read_reservoirs(which)[band], full.read_reservoirs(which)[band]), (rank, which)
        assert part.get_stats().rays[rr.RAY_GBUFFER] == rows.band_rows * W
        part.close()


def test_rccl_link_with_one_rank(atrium):
    """the built-in collectives with ONE rank (all a one-GPU box allows): librccl opened at run time, ncclCommInitRank, the in-place
    ncclAllGather on the reservoir stream after every spatial pass (moves nothing: results equal the plain context's), and the tile
    gather - uh_rccl_gather_tiles: pack, a grouped ncclSend / ncclRecv of the rank to itself, the root's one-launch composition,
    enqueued behind frames in flight with nothing waiting on the host - after which both images still equal the plain context's"""
    W, H = 96, 54
    plain = atrium.upload(rr.Renderer(W, H))
    linked = atrium.upload(rr.Renderer(W, H))
    linked.set_tile_partition(0, 1, 16)
    linked.rccl_attach(0, 1, rr.Renderer.rccl_unique_id())
    assert linked.rccl_comm_count() == 1
    loops = {}
    for r in (plain, linked):
        loops[r] = rr.FrameLoop(r, atrium.make_view(W, H))
        loops[r].frames(10, rr.PASS_ALL)
    # no synchronisation in between: the gather is ordered behind the wavefront by events, the frames after it behind the gather
    linked.rccl_gather_tiles(0, loops[linked].view.total_samples)
    for r in (plain, linked):
        loops[r].frames(3, rr.PASS_ALL)
    linked.rccl_gather_tiles(0, loops[linked].view.total_samples)
    assert np.array_equal(plain.read_accumulation().view(np.uint32), linked.read_accumulation().view(np.uint32))
    assert np.array_equal(plain.read_output_bgra8(), linked.read_output_bgra8())
    for which in range(3):
        assert same_reservoirs(plain.read_reservoirs(which), linked.read_reservoirs(which))
    # the communicator's rank and size are the partition's, or the call is refused
    linked.set_tile_partition(0, 2, 16)
    with pytest.raises(rr.UtopianError, match="communicator"):
        linked.rccl_gather_tiles(0, 1)
    linked.set_tile_partition(0, 1, 16)
    linked.rccl_detach()
    with pytest.raises(rr.UtopianError, match="no communicator"):
        linked.rccl_gather_tiles(0, 1)
    rr.FrameLoop(linked, atrium.make_view(W, H)).frames(2, rr.PASS_ALL)  # and goes on without the link
    linked.close()


def test_partition_change_keeps_the_history(atrium):
    """uh_set_restir_partition moves spatial_reuse_reservoirs (the temporal history) into the buffer of the new length"""
    W, H = 64, 50
    a = atrium.upload(rr.Renderer(W, H))
    run_frames(a, atrium, W, H, 3, rr.PASS_RESTIR)
    before = a.read_reservoirs(2).copy()
    a.set_restir_partition(1, 3)   # 3 x 17 rows: longer than the frame
    assert same_reservoirs(a.read_reservoirs(2), before)
    a.set_restir_partition(0, 1)
    assert same_reservoirs(a.read_reservoirs(2), before)
    b = atrium.upload(rr.Renderer(W, H))
    for r, n in ((a, 2), (b, 5)):
        loop = rr.FrameLoop(r, atrium.make_view(W, H))
        if r is a:
            loop.view.total_samples = 3  # the application's counter goes on (main.rs:467-469)
            loop.end_frame()             # ... and so does prev_frame_projection_view (main.rs:545-546)
        for _ in range(n):
            loop.frame(rr.PASS_RESTIR)
    assert same_reservoirs(a.read_reservoirs(2), b.read_reservoirs(2))


def test_error_paths_of_the_partition_verbs():
    import ctypes as C

    r = rr.Renderer(32, 24)
    with pytest.raises(rr.UtopianError):
        r.set_restir_partition(3, 3)       # rank < world
    with pytest.raises(rr.UtopianError):
        r.set_restir_partition(0, 0)
    lib = r._lib
    lib.uh_rccl_attach.argtypes, lib.uh_rccl_attach.restype = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p], C.c_int
    assert lib.uh_rccl_attach(r._ctx, 0, 1, None) != 0, "no id"
    assert lib.uh_rccl_attach(r._ctx, 2, 2, (C.c_uint8 * 128)()) != 0, "rank < world"
    lib.uh_rccl_detach.argtypes, lib.uh_rccl_detach.restype = [C.c_void_p], C.c_int
    assert lib.uh_rccl_detach(r._ctx) == 0, "detaching what was never attached is not an error"
    lib.uh_get_restir_rows.argtypes, lib.uh_get_restir_rows.restype = [C.c_void_p, C.c_void_p], C.c_int
    assert lib.uh_get_restir_rows(r._ctx, None) != 0
    # more ranks than rows: the ranks beyond the frame get an empty band and launch nothing
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    tiny = scene.upload(rr.Renderer(16, 3))
    tiny.set_restir_partition(5, 8)
    rows = tiny.restir_rows()
    assert rows.band_rows == 0 and rows.reuse_rows == 0 and rows.cast_rows == 0
    run_frames(tiny, scene, 16, 3, 2, rr.PASS_ALL, use_ris_light_sampling=1)
    assert tiny.get_stats().rays[rr.RAY_GBUFFER] == 0
    r.close()
    tiny.close()


def test_rank_that_owns_no_tile(atrium):
    """more ranks than tiles: a rank without pixels renders nothing and does not fall over (dense path ids: its slot is empty)"""
    W, H = 64, 64
    r = atrium.upload(rr.Renderer(W, H))
    r.set_tile_partition(5, 8, 64)  # one 64 x 64 tile in all: rank 0 owns it
    rr.FrameLoop(r, atrium.make_view(W, H)).frames(5, rr.PASS_ALL)
    s = r.get_stats()
    assert s.rays[rr.RAY_PRIMARY] == 0 and s.rays[rr.RAY_BOUNCE] == 0 and s.rays[rr.RAY_SUN_SHADOW] == 0
    assert not r.read_accumulation().any()
    owner = atrium.upload(rr.Renderer(W, H))
    owner.set_tile_partition(0, 8, 64)
    whole = atrium.upload(rr.Renderer(W, H))
    for x in (owner, whole):
        rr.FrameLoop(x, atrium.make_view(W, H)).frames(5, rr.PASS_ALL)
    assert np.array_equal(owner.read_accumulation().view(np.uint32), whole.read_accumulation().view(np.uint32))
