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
