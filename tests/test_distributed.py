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
