"""CPU: the N > 1 path with world_size 2 over gloo - each rank traces only its tiles (CPU oracle
backend), one gather composes the frame on rank 0, which must equal the single-rank frame bit for
bit (RNG is keyed on absolute pixel coordinates)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import oracle_api as oa
    import rust_renderer_amd as rr

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, tile = 80, 48, 16
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    o = scene.upload(oa.OracleRenderer(W, H, threads=2))
    o.set_tile_partition(rank, world, tile)
    loop = rr.FrameLoop(o, scene.make_view(W, H))
    for _ in range(2):
        loop.frame(rr.PASS_REFERENCE_PT)
    rays = torch.tensor([float(o.get_stats().path_rays)], dtype=torch.float64)
    dist.all_reduce(rays)
    composed = rr.distributed.gather_and_compose(o, rank, world, tile, dist, torch, "cpu")
    if rank == 0:
        np.save(out_path, composed)
        np.save(out_path + ".rays.npy", rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tile_partition_over_gloo(tmp_path):
    import torch.multiprocessing as mp

    import oracle_api as oa
    import rust_renderer_amd as rr

    out = str(tmp_path / "composed.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    composed = np.load(out)
    W, H = 80, 48
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    ref = scene.upload(oa.OracleRenderer(W, H))
    loop = rr.FrameLoop(ref, scene.make_view(W, H))
    for _ in range(2):
        loop.frame(rr.PASS_REFERENCE_PT)
    assert np.array_equal(composed.view(np.uint32), ref.read_accumulation().view(np.uint32))
    assert float(np.load(out + ".rays.npy")[0]) == float(ref.get_stats().path_rays), "the ranks' ray counts add up to the full frame's"


def _launch_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("uh_launch", os.path.join(ROOT, "rust-renderer_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.timeout(300)
def test_launcher_runs_two_ranks(tmp_path):
    """the launcher bench.py uses for a plain `--gpus N` (rust-renderer_amd/launch.py: N child processes under
    torch.distributed.run on 127.0.0.1) runs a 2-rank gloo job whose composed frame equals the single-rank one"""
    import oracle_api as oa
    import rust_renderer_amd as rr

    launch = _launch_module()
    out = str(tmp_path / "composed.npy")
    rc = launch.spawn_ranks(2, os.path.join(ROOT, "tests", "dist_worker.py"), [out], timeout=280)
    assert rc == 0
    W, H = 80, 48
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    ref = scene.upload(oa.OracleRenderer(W, H))
    loop = rr.FrameLoop(ref, scene.make_view(W, H))
    for _ in range(2):
        loop.frame(rr.PASS_REFERENCE_PT)
    assert np.array_equal(np.load(out).view(np.uint32), ref.read_accumulation().view(np.uint32))
    assert launch.spawn_ranks(2, sys.executable, ["-c", "raise SystemExit(3)"], timeout=120) != 0, "a failing rank must fail the job"


@pytest.mark.timeout(300)
def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher around it must start two ranks that reach the process group
    (it used to sys.exit). On this GPU-less box each rank then stops at uh_create: NO_DEVICE - there is no CPU
    fallback in the product path - and the parent hands that failure on."""
    import subprocess

    import rust_renderer_amd as rr

    lib = rr.load_library()
    ctx = __import__("ctypes").c_void_p()
    if lib.uh_create(0, 16, 16, __import__("ctypes").byref(ctx)) == 0:
        lib.uh_destroy(ctx)
        pytest.skip("a GPU is visible: the 2-rank run is the GPU box's job")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert p.returncode != 0
    assert "rank 0/2 joined the gloo group" in p.stderr and "rank 1/2 joined the gloo group" in p.stderr, p.stderr[-2000:]
    assert "NO_DEVICE" in p.stderr


def _restir_worker(rank, world, port, out_path):
    """config-2 style job on `world` ranks: tiles for the path tracer, bands of rows for the reservoir passes with one
    all-gather of spatial_reuse_reservoirs per frame (rust-renderer_amd/distributed.py partition_reservoir_passes)"""
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import oracle_api as oa
    import rust_renderer_amd as rr

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, tile = 72, 50, 16  # 50 rows over 3 ranks: bands of 17, 17, 16; the first band also needs the last row
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    o = scene.upload(oa.OracleRenderer(W, H, threads=2))
    o.set_tile_partition(rank, world, tile)
    rr.distributed.partition_reservoir_passes(o, rank, world, dist, torch)
    loop = rr.FrameLoop(o, scene.make_view(W, H, use_ris_light_sampling=1))
    for _ in range(3):
        loop.frame(rr.PASS_ALL)
    rays = torch.tensor([float(x) for x in o.get_stats().rays], dtype=torch.float64)
    dist.all_reduce(rays)
    spatial = o.read_reservoirs(2)  # whole frame on every rank
    initial = rr.distributed.gather_reservoir_rows(o, 0, rank, world, dist, torch)
    temporal = rr.distributed.gather_reservoir_rows(o, 1, rank, world, dist, torch)
    composed = rr.distributed.gather_and_compose(o, rank, world, tile, dist, torch, "cpu")
    np.savez(out_path + f".rank{rank}.npz", spatial=spatial, initial=initial, temporal=temporal, rays=rays.numpy(),
             composed=composed if rank == 0 else np.zeros(0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_three_rank_reservoir_band_partition_over_gloo(tmp_path):
    """every rank ends every frame with the single-rank spatial_reuse_reservoirs, bit for bit; the initial and temporal
    buffers assemble from the bands; the composed frame and the ray counts equal the single-rank run's"""
    import torch.multiprocessing as mp

    import oracle_api as oa
    import rust_renderer_amd as rr

    world = 3
    out = str(tmp_path / "restir")
    mp.spawn(_restir_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    W, H = 72, 50
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    ref = scene.upload(oa.OracleRenderer(W, H))
    loop = rr.FrameLoop(ref, scene.make_view(W, H, use_ris_light_sampling=1))
    for _ in range(3):
        loop.frame(rr.PASS_ALL)
    assert (ref.read_reservoirs(2)["M"] > 1).any(), "the history builds up: frame 3's temporal pass did read frame 2's all-gathered buffer"
    for rank in range(world):
        got = np.load(out + f".rank{rank}.npz")
        for which, name in ((2, "spatial"), (0, "initial"), (1, "temporal")):
            assert np.array_equal(got[name].view(np.uint8), ref.read_reservoirs(which).view(np.uint8)), (rank, name)
        assert list(got["rays"]) == [float(x) for x in ref.get_stats().rays], "G-buffer rays are counted once per pixel, by the band's owner"
    assert np.array_equal(np.load(out + ".rank0.npz")["composed"].view(np.uint32), ref.read_accumulation().view(np.uint32))


def test_reservoir_rows_cover_what_the_passes_read():
    """the rows a rank computes (orc_get_restir_rows, the checker's definitional restatement): bands tile the frame; the reuse rows
    hold every row a band's spatial pass can gather from (|dy| < 30, negative rows wrap to the last one); the cast rows hold the
    row above each reuse row"""
    import oracle_api as oa

    for H, world in ((50, 3), (1080, 8), (64, 4), (31, 2), (7, 8), (2160, 8), (100, 1)):
        covered = np.zeros(H, dtype=int)
        for rank in range(world):
            o = oa.OracleRenderer(4, H)
            o.set_restir_partition(rank, world)
            r = o.restir_rows()
            covered[r.band_row0:r.band_row0 + r.band_rows] += 1
            if r.band_rows == 0:
                continue
            reuse = set(range(r.reuse_row0, r.reuse_row0 + r.reuse_rows)) | set(range(r.reuse_extra_row0, r.reuse_extra_row0 + r.reuse_extra_rows))
            cast = set(range(r.cast_row0, r.cast_row0 + r.cast_rows)) | set(range(r.cast_extra_row0, r.cast_extra_row0 + r.cast_extra_rows))
            for y in range(r.band_row0, r.band_row0 + r.band_rows):
                for dy in range(-29, 30):
                    ny = y + dy
                    assert (H - 1 if ny < 0 or ny > H - 1 else ny) in reuse, (H, world, rank, y, dy)
            assert all(y in cast and max(y - 1, 0) in cast for y in reuse)
            if world == 1:
                assert (r.band_rows, r.reuse_rows, r.cast_rows) == (H, H, H)
        assert (covered == 1).all(), (H, world)


class _MailboxGather:
    """torch.distributed.gather for ranks that live in ONE process: the non-root ranks' calls leave their device tensor in a
    shared mailbox, the root's call (made last) copies them into its gather list - device to device, as RCCL would deliver them"""

    def __init__(self, rank, mailbox):
        self.rank, self.mailbox = rank, mailbox

    def gather(self, tensor, gather_list=None, dst=0):
        if self.rank != dst:
            assert gather_list is None
            self.mailbox[self.rank] = tensor.clone()
            return
        for r, slot in enumerate(gather_list):
            slot.copy_(tensor if r == self.rank else self.mailbox[r])


@pytest.mark.gpu
@pytest.mark.parametrize("resolve", [True, False])
def test_device_branch_of_the_composition_with_three_ranks_in_one_process(resolve):
    """the HIP branch of distributed.gather_and_compose (uh_pack_tiles -> gather of DEVICE tensors ->
    uh_compose_tiles / uh_unpack_tiles on the root) had only ever run at world = 1. Three contexts on GPU 0 stand for three
    ranks; the tensors travel through a mailbox in place of the RCCL gather; the root's image must be the single context's."""
    torch = pytest.importorskip("torch")
    import rust_renderer_amd as rr

    W, H, tile, world, frames = 200, 120, 32, 3, 5  # 7 x 4 tiles, the last column and row partial; 28 tiles over 3 ranks: 10 / 9 / 9
    scene = rr.scenes.cornell_scene(subdivisions=2, tex_size=16)
    single = scene.upload(rr.Renderer(W, H, device=0))
    loop = rr.FrameLoop(single, scene.make_view(W, H))
    loop.frames(frames, rr.PASS_REFERENCE_PT)
    want_acc, want_out, want_rays = single.read_accumulation(), single.read_output_bgra8(), single.get_stats().path_rays
    total = loop.view.total_samples
    ranks, mailbox, rays = [], {}, 0
    for r in range(world):
        ctx = scene.upload(rr.Renderer(W, H, device=0))
        ctx.set_tile_partition(r, world, tile)
        lp = rr.FrameLoop(ctx, scene.make_view(W, H))
        lp.frames(frames, rr.PASS_REFERENCE_PT)
        rays += ctx.get_stats().path_rays
        ranks.append(ctx)
    assert rays == want_rays, "the ranks' ray counts add up to the full frame's"
    for r in (2, 1, 0):  # the root last: its gather finds the others' tiles in the mailbox
        rr.distributed.gather_and_compose(ranks[r], r, world, tile, _MailboxGather(r, mailbox), torch, "cuda:0",
                                          resolve=(total, loop.view.accumulation_limit) if resolve else None)
    got = ranks[0].read_accumulation()
    assert np.array_equal(got.view(np.uint32), want_acc.view(np.uint32))
    if resolve:
        assert np.array_equal(ranks[0].read_output_bgra8(), want_out)  # uh_compose_tiles recomputed pt_output_image for every pixel
    else:
        ranks[0].resolve_output(total, loop.view.accumulation_limit)
        assert np.array_equal(ranks[0].read_output_bgra8(), want_out)
