"""CPU: the N > 1 path with 2 and 3 ranks - each rank is a process of its own that traces only its tiles (CPU oracle backend)
and finds the others through the launcher's rendezvous (rust-renderer_amd/launch.py: TCP on 127.0.0.1, no torch); one gather
composes the frame on rank 0, which must equal the single-rank frame bit for bit (RNG is keyed on absolute pixel coordinates,
SURVEY.md section 8e). On GPUs the same gather is RCCL inside the library (uh_rccl_gather_tiles; tests/test_gpu_restir_partition.py
rehearses it with one rank, tests/test_gpu_parity.py holds the device branch of the composition with three contexts)."""
import multiprocessing as mp
import os
import sys
import uuid

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(target, world, *args):
    """`world` fresh processes (spawn, not fork: each loads the libraries itself); every one must exit 0"""
    ctx = mp.get_context("spawn")
    key = "test_" + uuid.uuid4().hex
    procs = [ctx.Process(target=target, args=(r, world, key) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    assert [p.exitcode for p in procs] == [0] * world


def _paths():
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _worker(rank, world, key, out_path):
    _paths()
    import oracle_api as oa
    import rust_renderer_amd as rr

    rdzv = rr.launch.Rendezvous(rank, world, key)
    W, H, tile = 80, 48, 16
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    o = scene.upload(oa.OracleRenderer(W, H, threads=2))
    o.set_tile_partition(rank, world, tile)
    loop = rr.FrameLoop(o, scene.make_view(W, H))
    for _ in range(2):
        loop.frame(rr.PASS_REFERENCE_PT)
    rays = rdzv.allreduce([float(o.get_stats().path_rays)], "sum")
    composed = rr.distributed.gather_and_compose(o, rdzv, tile)
    if rank == 0:
        np.save(out_path, composed)
        np.save(out_path + ".rays.npy", np.float64(rays))
    rdzv.barrier()
    rdzv.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_tile_partition_over_the_rendezvous(tmp_path, world):
    import oracle_api as oa
    import rust_renderer_amd as rr

    out = str(tmp_path / "composed.npy")
    _spawn(_worker, world, out)
    composed = np.load(out)
    W, H = 80, 48
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    ref = scene.upload(oa.OracleRenderer(W, H))
    loop = rr.FrameLoop(ref, scene.make_view(W, H))
    for _ in range(2):
        loop.frame(rr.PASS_REFERENCE_PT)
    assert np.array_equal(composed.view(np.uint32), ref.read_accumulation().view(np.uint32))
    assert float(np.load(out + ".rays.npy")[0]) == float(ref.get_stats().path_rays), "the ranks' ray counts add up to the full frame's"


def _rdzv_worker(rank, world, key, out_path):
    _paths()
    import rust_renderer_amd as rr

    rdzv = rr.launch.Rendezvous(rank, world, key)
    got = {
        "bcast": rdzv.broadcast(bytes(range(128)) if rank == 0 else None),
        "bcast_from_last": rdzv.broadcast(b"x" * (1 << 20) if rank == world - 1 else None, src=world - 1),
        "all": rdzv.allgather(bytes([rank]) * (rank + 1)),
        "gathered": rdzv.gather(bytes([rank]), dst=1),
        "sum": rdzv.allreduce([rank + 0.5, 1.0], "sum"),
        "max": rdzv.allreduce([float(rank), -float(rank)], "max"),
    }
    rdzv.barrier()
    rdzv.close()
    assert got["bcast"] == bytes(range(128)) and got["bcast_from_last"] == b"x" * (1 << 20)
    assert got["all"] == [bytes([r]) * (r + 1) for r in range(world)]
    assert got["gathered"] == ([bytes([r]) for r in range(world)] if rank == 1 else None)
    assert got["sum"] == [sum(r + 0.5 for r in range(world)), float(world)] and got["max"] == [float(world - 1), 0.0]
    open(out_path + f".{rank}", "w").write("ok")


@pytest.mark.timeout(120)
def test_rendezvous_collectives_with_four_ranks(tmp_path):
    """what a GPU rank uses instead of torch.distributed: the id broadcast, the barrier and the reductions around the timed region"""
    out = str(tmp_path / "rdzv")
    _spawn(_rdzv_worker, 4, out)
    assert all(os.path.exists(out + f".{r}") for r in range(4))


def test_rendezvous_ignores_a_stale_port_file(tmp_path):
    """a file an earlier job of the same key left behind (dead port) must not keep the ranks apart"""
    import tempfile

    import rust_renderer_amd as rr

    key = "stale_" + uuid.uuid4().hex
    with open(os.path.join(tempfile.gettempdir(), f"utopian_rdzv_{key}"), "w") as f:
        f.write(str(rr.launch.free_port()))  # nobody listens there
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_stale_worker, args=(r, 2, key)) for r in (1, 0)]  # rank 1 first: it meets the stale file
    procs[0].start()
    import time

    time.sleep(0.5)
    procs[1].start()
    for p in procs:
        p.join(60)
    assert [p.exitcode for p in procs] == [0, 0]


def _stale_worker(rank, world, key):
    _paths()
    import rust_renderer_amd as rr

    rdzv = rr.launch.Rendezvous(rank, world, key, timeout=30)
    assert rdzv.allgather(bytes([rank])) == [b"\x00", b"\x01"]
    rdzv.close()


def _launch_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("uh_launch", os.path.join(ROOT, "rust-renderer_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.timeout(300)
def test_launcher_runs_two_ranks(tmp_path):
    """the launcher bench.py uses for a plain `--gpus N` (rust-renderer_amd/launch.py: N child processes under
    torch.distributed.run on 127.0.0.1) runs a 2-rank job - the ranks torch-free, meeting through launch.Rendezvous.from_env() -
    whose composed frame equals the single-rank one"""
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
    """`python bench.py --gpus 2` without a launcher around it must start two ranks that find each other
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
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert p.returncode != 0
    assert "rank 0/2 joined the rendezvous" in p.stderr and "rank 1/2 joined the rendezvous" in p.stderr, p.stderr[-2000:]
    assert "NO_DEVICE" in p.stderr


def _restir_worker(rank, world, key, out_path):
    """config-2 style job on `world` ranks: tiles for the path tracer, bands of rows for the reservoir passes with one
    all-gather of spatial_reuse_reservoirs per frame (rust-renderer_amd/distributed.py partition_reservoir_passes)"""
    _paths()
    import oracle_api as oa
    import rust_renderer_amd as rr

    rdzv = rr.launch.Rendezvous(rank, world, key)
    W, H, tile = 72, 50, 16  # 50 rows over 3 ranks: bands of 17, 17, 16; the first band also needs the last row
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    o = scene.upload(oa.OracleRenderer(W, H, threads=2))
    o.set_tile_partition(rank, world, tile)
    rr.distributed.partition_reservoir_passes(o, rdzv)
    loop = rr.FrameLoop(o, scene.make_view(W, H, use_ris_light_sampling=1))
    for _ in range(3):
        loop.frame(rr.PASS_ALL)
    rays = rdzv.allreduce([float(x) for x in o.get_stats().rays], "sum")
    spatial = o.read_reservoirs(2)  # whole frame on every rank
    initial = rr.distributed.gather_reservoir_rows(o, 0, rdzv)
    temporal = rr.distributed.gather_reservoir_rows(o, 1, rdzv)
    composed = rr.distributed.gather_and_compose(o, rdzv, tile)
    np.savez(out_path + f".rank{rank}.npz", spatial=spatial, initial=initial, temporal=temporal, rays=np.float64(rays),
             composed=composed if rank == 0 else np.zeros(0))
    rdzv.barrier()
    rdzv.close()


@pytest.mark.timeout(600)
def test_three_rank_reservoir_band_partition(tmp_path):
    """every rank ends every frame with the single-rank spatial_reuse_reservoirs, bit for bit; the initial and temporal
    buffers assemble from the bands; the composed frame and the ray counts equal the single-rank run's"""
    import oracle_api as oa
    import rust_renderer_amd as rr

    world = 3
    out = str(tmp_path / "restir")
    _spawn(_restir_worker, world, out)
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
