"""Rank script for tests/test_gpu_rehearsal.py: one of N rank PROCESSES of a tile-partitioned job, all on GPU 0, with tests/cpp/fake_rccl.cpp
standing in for librccl (found first on LD_LIBRARY_PATH). What a rank of `bench.py --gpus N` does: rendezvous (no torch), attach
(ncclCommInitRank; the reservoir passes by bands of rows with the all-gather inside the library), its tiles of every frame, one
uh_rccl_gather_tiles per composed frame. Rank 0 saves the composed images and the spatial reservoirs to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import numpy as np

    import rust_renderer_amd as rr

    out, W, H, tile, frames, lights = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    rdzv = rr.launch.Rendezvous.from_env()
    rank, world = rdzv.rank, rdzv.world
    scene = rr.scenes.sponza_class_scene(detail=0.12, tex_size=32, with_spheres=True, num_lights=64 if lights else 0, sphere_subdivisions=2)
    r = scene.upload(rr.Renderer(W, H, device=0))  # every rank on the one GPU of the box
    r.set_tile_partition(rank, world, tile)
    rr.distributed.attach_ranks(r, rdzv)
    assert r.rccl_comm_count() == world
    mask = rr.PASS_ALL if lights else rr.PASS_REFERENCE_PT
    loop = rr.FrameLoop(r, scene.make_view(W, H, use_ris_light_sampling=1 if lights else 0))
    for _ in range(frames):
        loop.frame(mask)
        rr.distributed.gather_and_compose(r, rdzv, tile, resolve=(loop.view.total_samples, loop.view.accumulation_limit))
    r.synchronize()
    if rank == 0:
        np.savez(out, acc=r.read_accumulation(), bgra=r.read_output_bgra8(), spatial=r.read_reservoirs(2), rays=np.array(list(r.get_stats().rays), dtype=np.int64))
    rays = rdzv.allreduce([int(x) for x in r.get_stats().rays], "sum")
    if rank == 0:
        np.save(out + ".rays.npy", np.array(rays, dtype=np.int64))
    rdzv.barrier()
    r.rccl_detach()
    rdzv.close()
    assert "torch" not in sys.modules, "a rank imports no torch"


if __name__ == "__main__":
    main()
