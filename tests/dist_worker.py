"""Rank script for tests/test_distributed.py::test_launcher_runs_two_ranks: started by rust-renderer_amd/launch.py
(the launcher bench.py uses for --gpus N); finds the other ranks through launch.Rendezvous.from_env() like a GPU rank does (no
torch), traces its tiles with the CPU oracle and takes part in the one composition gather. Rank 0 saves the composed frame to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import numpy as np

    import oracle_api as oa
    import rust_renderer_amd as rr

    rdzv = rr.launch.Rendezvous.from_env()
    rank, world = rdzv.rank, rdzv.world
    assert (rank, world) == (int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])) and "torch" not in sys.modules
    W, H, tile = 80, 48, 16
    scene = rr.scenes.cornell_scene(subdivisions=1, tex_size=8)
    o = scene.upload(oa.OracleRenderer(W, H, threads=2))
    o.set_tile_partition(rank, world, tile)
    loop = rr.FrameLoop(o, scene.make_view(W, H))
    for _ in range(2):
        loop.frame(rr.PASS_REFERENCE_PT)
    composed = rr.distributed.gather_and_compose(o, rdzv, tile)
    if rank == 0:
        np.save(sys.argv[1], composed)
    rdzv.barrier()
    rdzv.close()
    assert "torch" not in sys.modules, "a rank imports no torch"


if __name__ == "__main__":
    main()
