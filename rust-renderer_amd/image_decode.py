"""In-repo image decoding for glTF ingestion: PNG is decoded here (zlib + the five scanline
filters, numpy), with the reference loader's format policy on top: RGB8 is expanded to RGBA8 with alpha 255, RGBA8 passes,
every other pixel format is the reference's `panic!("Unsupported image format!")` (utopian/src/gltf_loader.rs:179-198; the
`gltf` crate hands 8-bit grey / grey-alpha / 16-bit images over as R8, R8G8, R16...). JPEG (65 of Sponza's 69 images) is decoded
by jpeg_decode.py, the twin of include/utopian_jpeg.hpp.
"""
import struct
import zlib

import numpy as np

PNG_MAGIC = b"\x89PNG\r\n\x1a\n"


class UnsupportedImage(ValueError):
    pass


def _unfilter(raw, height, stride, bpp):
    """PNG scanline filters 0-4 (None, Sub, Up, Average, Paeth) over `height` rows of `stride` bytes"""
    out = np.zeros((height, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    pos = 0
    for y in range(height):
        ft = raw[pos]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=pos + 1).astype(np.int32)
        pos += stride + 1
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.copy()
            # Sub: each byte adds the byte bpp to its left: a running sum per byte lane
            for c in range(bpp):
                cur[c::bpp] = np.cumsum(line[c::bpp]) & 255
        elif ft in (3, 4):
            cur = np.zeros(stride, dtype=np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    c = prev[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        else:
            raise UnsupportedImage(f"PNG filter type {ft}")
        out[y] = cur
        prev = cur
    return out


def decode_png(data):
    """-> (pixels (H, W, C) uint8 or uint16, format name as the gltf crate reports it: R8 / R8G8 / R8G8B8 / R8G8B8A8 / R16...)"""
    if data[:8] != PNG_MAGIC:
        raise UnsupportedImage("not a PNG")
    pos = 8
    ihdr = None
    idat = []
    palette = trns = None
    while pos < len(data):
        (length,), kind = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if kind == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            palette = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = body
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = ihdr
    if interlace:
        raise UnsupportedImage("interlaced PNG")
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    if depth not in (8, 16) and not (ctype in (0, 3) and depth in (1, 2, 4)):
        raise UnsupportedImage(f"PNG bit depth {depth}")
    bits = channels * depth
    stride = (w * bits + 7) // 8
    rows = _unfilter(zlib.decompress(b"".join(idat)), h, stride, max(1, bits // 8))
    if depth < 8:  # packed grey / palette indices
        px = np.unpackbits(rows, axis=1)[:, : w * depth].reshape(h, w, depth)
        vals = np.zeros((h, w), dtype=np.uint8)
        for b in range(depth):
            vals = (vals << 1) | px[:, :, b]
        rows8 = vals if ctype == 3 else (vals.astype(np.uint16) * (255 // ((1 << depth) - 1))).astype(np.uint8)
        img = rows8[:, :, None]
    elif depth == 8:
        img = rows.reshape(h, w, channels)
    else:
        img = rows.reshape(h, w, channels, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]
    if ctype == 3:  # palette -> RGB8 (RGBA8 with a tRNS chunk), as the image crate expands it
        idx = img[:, :, 0]
        rgb = palette[idx]
        if trns is not None:
            alpha = np.full(256, 255, dtype=np.uint8)
            alpha[: len(trns)] = np.frombuffer(trns, dtype=np.uint8)
            return np.concatenate([rgb, alpha[idx][:, :, None]], axis=2), "R8G8B8A8"
        return rgb, "R8G8B8"
    name = {1: "R", 2: "RG", 3: "RGB", 4: "RGBA"}[channels]
    return img, "".join(f"{c}{depth}" for c in name)


def load_image_rgba8(data):
    """bytes of an image file -> (H, W, 4) uint8 with the reference loader's policy (gltf_loader.rs:179-198)"""
    if data[:8] == PNG_MAGIC:
        img, fmt = decode_png(data)
    elif data[:2] == b"\xff\xd8":
        from .jpeg_decode import JpegError, decode_jpeg

        try:
            img, _ = decode_jpeg(data)
        except JpegError as e:
            raise UnsupportedImage(f"JPEG: {e}") from e
        fmt = "R8" if img.shape[2] == 1 else "R8G8B8"  # the image crate reports a grey JPEG as L8
    else:
        raise UnsupportedImage("neither PNG nor JPEG")
    if fmt == "R8G8B8":  # "Convert images from rgb8 to rgba8"
        return np.concatenate([img, np.full(img.shape[:2] + (1,), 255, dtype=np.uint8)], axis=2)
    if fmt != "R8G8B8A8":
        raise UnsupportedImage(f"Unsupported image format! ({fmt}; the reference loader panics on anything but RGB8 / RGBA8)")
    return np.ascontiguousarray(img)
