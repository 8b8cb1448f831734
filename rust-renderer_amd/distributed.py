"""Multi-GPU composition: the framebuffer is partitioned into tile_size x tile_size tiles owned
round-robin (tile id % world == rank); every rank path-traces only its tiles and ONE gather of the
packed RGBA32F accumulation tiles to rank 0 composes the frame. The reference is single-device
(utopian/src/device.rs:45); this is the build's own addition.

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


def gather_and_compose(renderer, rank, world, tile, dist, torch, device, resolve=None):
    """One collective per composed frame. HIP renderers pack on the device and hand RCCL a device
    buffer; CPU (oracle / gloo) renderers go through the numpy restatement. Returns, on rank 0, the
    composed (H, W, 4) accumulation as numpy for CPU renderers, or None after composing in place
    on the device for HIP renderers. `resolve` = (total_samples, accumulation_limit): the root also recomputes
    pt_output_image, in the same launch that scatters the tiles (uh_compose_tiles)."""
    counts = [c * tile * tile for c in tile_counts(renderer.width, renderer.height, tile, world)]
    if renderer.backend == "hip":
        buf = torch.empty((max(counts), 4), dtype=torch.float32, device=device)
        renderer.pack_tiles(buf.data_ptr(), buf.shape[0])
        every = torch.empty((world, max(counts), 4), dtype=torch.float32, device=device) if rank == 0 else None
        dist.gather(buf, list(every.unbind(0)) if rank == 0 else None, dst=0)
        if rank == 0:
            torch.cuda.synchronize()
            if resolve is not None:
                renderer.compose_tiles(every.data_ptr(), max(counts), resolve[0], resolve[1])
            else:
                for r in range(1, world):
                    renderer.unpack_tiles(r, every[r].data_ptr(), counts[r])
        return None
    acc = renderer.read_accumulation()
    buf = torch.zeros((max(counts), 4), dtype=torch.float32)
    buf[: counts[rank]] = torch.from_numpy(pack_tiles_host(acc, tile, rank, world))
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, parts, dst=0)
    if rank == 0:
        for r in range(1, world):
            unpack_tiles_host(acc, parts[r].numpy(), tile, r, world)
        return acc
    return None


def partition_reservoir_passes(renderer, rank, world, dist, torch):
    """The G-buffer cast and the reservoir passes of `renderer` cover rank's band of rows from now on, and every spatial
    pass is followed by an all-gather of the bands, so that every rank holds the whole spatial_reuse_reservoirs of every
    frame (temporal_reuse.rgen:90-99 reads it at a reprojected pixel, the path tracer at its own tiles).
    HIP renderers: RCCL inside the library, on the stream the passes run on (uh_rccl_attach; the 128-byte id travels through
    ONE torch.distributed broadcast, then torch is out of the loop). CPU renderers (the oracle under gloo, in the CPU tests):
    the same library hook calls back into Python, synchronously, and gloo moves the bands."""
    if world <= 1:
        renderer.set_restir_partition(0, 1)
        return
    if renderer.backend == "hip":
        ident = [renderer.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        renderer.rccl_attach(rank, world, ident[0])
        return
    import ctypes as C

    def exchange(stream, base, band_bytes, r, w):
        whole = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_uint8)), shape=(w * band_bytes,))
        mine = torch.from_numpy(whole[r * band_bytes:(r + 1) * band_bytes].copy())
        parts = [torch.empty_like(mine) for _ in range(w)]
        dist.all_gather(parts, mine)
        for k in range(w):
            if k != r:
                whole[k * band_bytes:(k + 1) * band_bytes] = parts[k].numpy()
        return 0

    renderer.set_restir_partition(rank, world, exchange)


def gather_reservoir_rows(renderer, which, rank, world, dist, torch):
    """initial (0) / temporal (1) reservoirs of the whole frame on every rank, from the bands the ranks computed (tests)."""
    rows = renderer.restir_rows()
    mine = renderer.read_reservoirs(which)
    B, W = rows.rows_per_band, renderer.width
    send = np.zeros((B, W), dtype=mine.dtype)
    send[: rows.band_rows] = mine[rows.band_row0:rows.band_row0 + rows.band_rows]
    t = torch.from_numpy(send.view(np.uint8).reshape(-1).copy())
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    out = np.concatenate([p.numpy().view(mine.dtype).reshape(B, W) for p in parts])[: renderer.height]
    return out
