"""Rank script for tests/test_distributed.py::test_launcher_runs_two_ranks: started by rust-renderer_amd/launch.py
(the launcher bench.py uses for --gpus N) as one rank of a gloo group; traces its tiles with the CPU oracle
and takes part in the one composition gather. Rank 0 saves the composed frame to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    import oracle_api as oa
    import rust_renderer_amd as rr

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, tile = 80, 48, 16
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    o = scene.upload(oa.OracleRenderer(W, H, threads=2))
    o.set_tile_partition(rank, world, tile)
    loop = rr.FrameLoop(o, scene.make_view(W, H))
    for _ in range(2):
        loop.frame(rr.PASS_REFERENCE_PT)
    composed = rr.distributed.gather_and_compose(o, rank, world, tile, dist, torch, "cpu")
    if rank == 0:
        np.save(sys.argv[1], composed)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
