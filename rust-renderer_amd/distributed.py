"""Multi-GPU composition: the framebuffer is partitioned into tile_size x tile_size tiles owned
round-robin (tile id % world == rank); every rank path-traces only its tiles and ONE gather of the
packed RGBA32F accumulation tiles to rank 0 composes the frame. The reference is single-device
(utopian/src/device.rs:45); this is the build's own addition.
On GPUs the gather is RCCL INSIDE the library (uh_rccl_gather_tiles); what is left here is the job-level glue - handing the
ncclUniqueId round through the launcher's rendezvous (launch.Rendezvous: TCP on 127.0.0.1, no torch in a GPU process) - and the
numpy twin of the index math for the CPU oracle.

The reservoir passes (config 2) are partitioned differently - by bands of rows, with ONE all-gather of
spatial_reuse_reservoirs per frame (partition_reservoir_passes below; include/utopian_hip.h uh_set_restir_partition).

The same index math runs in three places and is tested against each other: k_tiles (HIP,
csrc/kernels.hip), pack_tiles_host / unpack_tiles_host here (numpy; CPU gloo tests and the oracle),
and owns_pixel in both tracers.
"""
import numpy as np


def tile_counts(W, H, tile, world):
    tiles = -(-W // tile) * -(-H // tile)
    return [(tiles - r + world - 1) // world if tiles > r else 0 for r in range(world)]


def tile_pixel_index(W, H, tile, rank, world):
    """(packed_count, flat image index per packed slot or -1 for padding outside the image)."""
    tiles_x = -(-W // tile)
    owned = tile_counts(W, H, tile, world)[rank]
    t = rank + np.arange(owned, dtype=np.int64) * world
    tx, ty = (t % tiles_x) * tile, (t // tiles_x) * tile
    wy, wx = np.divmod(np.arange(tile * tile, dtype=np.int64), tile)
    x = tx[:, None] + wx[None, :]
    y = ty[:, None] + wy[None, :]
    idx = np.where((x < W) & (y < H), y * W + x, -1)
    return owned * tile * tile, idx.reshape(-1)


def owner_map(W, H, tile, world):
    tiles_x = -(-W // tile)
    y, x = np.mgrid[0:H, 0:W]
    return ((y // tile) * tiles_x + (x // tile)) % world


def pack_tiles_host(acc, tile, rank, world):
    H, W = acc.shape[:2]
    n, idx = tile_pixel_index(W, H, tile, rank, world)
    out = np.zeros((n, 4), dtype=np.float32)
    valid = idx >= 0
    out[valid] = acc.reshape(-1, 4)[idx[valid]]
    return out


def unpack_tiles_host(acc, packed, tile, rank, world):
    H, W = acc.shape[:2]
    _, idx = tile_pixel_index(W, H, tile, rank, world)
    valid = idx >= 0
    acc.reshape(-1, 4)[idx[valid]] = packed[: len(idx)][valid]
    return acc


def gather_and_compose(renderer, rdzv, tile, resolve=None, root=0):
    """One collective per composed frame. `rdzv`: the job's launch.Rendezvous (rank, world, and - for CPU renderers - the transport).
    HIP renderers: uh_rccl_gather_tiles - pack, grouped ncclSend / ncclRecv over the communicator uh_rccl_attach made, composition on
    the root, all enqueued on the context's stream inside the library (no torch, no host wait; `resolve` = (total_samples,
    accumulation_limit) is what the root recomputes pt_output_image with). Returns None: the root's images hold the frame.
    CPU renderers (the oracle in the CPU tests): the numpy restatement of the same pack / unpack index math, the tiles travelling
    through the rendezvous' sockets; returns the composed (H, W, 4) accumulation on the root, None elsewhere."""
    rank, world = rdzv.rank, rdzv.world
    if renderer.backend == "hip":
        total, limit = resolve if resolve is not None else (1, 999999)
        renderer.rccl_gather_tiles(root, total, limit)
        return None
    acc = renderer.read_accumulation()
    parts = rdzv.gather(pack_tiles_host(acc, tile, rank, world).tobytes(), dst=root)
    if rank != root:
        return None
    for r in range(world):
        if r != root:
            unpack_tiles_host(acc, np.frombuffer(parts[r], dtype=np.float32).reshape(-1, 4), tile, r, world)
    return acc


def attach_ranks(renderer, rdzv, band_partition=True):
    """HIP renderers: rank 0 makes the ncclUniqueId, the rendezvous hands its 128 bytes to every rank, every rank attaches
    (uh_rccl_attach: ncclCommInitRank + the reservoir passes by bands of rows with an in-library all-gather per frame).
    band_partition = False: full-frame reservoir passes on every rank (rounds 1-2), the communicator stays for the tile gather."""
    ident = rdzv.broadcast(renderer.rccl_unique_id() if rdzv.rank == 0 else None)
    renderer.rccl_attach(rdzv.rank, rdzv.world, ident)
    if not band_partition:
        renderer.set_restir_partition(0, 1)


def partition_reservoir_passes(renderer, rdzv):
    """The G-buffer cast and the reservoir passes of `renderer` cover rank's band of rows from now on, and every spatial
    pass is followed by an all-gather of the bands, so that every rank holds the whole spatial_reuse_reservoirs of every
    frame (temporal_reuse.rgen:90-99 reads it at a reprojected pixel, the path tracer at its own tiles).
    HIP renderers: RCCL inside the library, on the stream the passes run on (attach_ranks). CPU renderers (the oracle, in the CPU
    tests): the same library hook calls back into Python, synchronously, and the rendezvous' sockets move the bands."""
    rank, world = rdzv.rank, rdzv.world
    if world <= 1:
        renderer.set_restir_partition(0, 1)
        return
    if renderer.backend == "hip":
        attach_ranks(renderer, rdzv)
        return
    import ctypes as C

    def exchange(stream, base, band_bytes, r, w):
        whole = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_uint8)), shape=(w * band_bytes,))
        parts = rdzv.allgather(whole[r * band_bytes:(r + 1) * band_bytes].tobytes())
        for k in range(w):
            if k != r:
                whole[k * band_bytes:(k + 1) * band_bytes] = np.frombuffer(parts[k], dtype=np.uint8)
        return 0

    renderer.set_restir_partition(rank, world, exchange)


def gather_reservoir_rows(renderer, which, rdzv):
    """initial (0) / temporal (1) reservoirs of the whole frame on every rank, from the bands the ranks computed (tests)."""
    rows = renderer.restir_rows()
    mine = renderer.read_reservoirs(which)
    B, W = rows.rows_per_band, renderer.width
    send = np.zeros((B, W), dtype=mine.dtype)
    send[: rows.band_rows] = mine[rows.band_row0:rows.band_row0 + rows.band_rows]
    parts = rdzv.allgather(send.tobytes())
    out = np.concatenate([np.frombuffer(p, dtype=mine.dtype).reshape(B, W) for p in parts])[: renderer.height]
    return out
