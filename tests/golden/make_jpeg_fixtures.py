#!/usr/bin/env python3
"""Makes tests/golden/jpeg_fixtures.npz: inputs and expected outputs for the two in-repo JPEG decoders (include/utopian_jpeg.hpp,
rust-renderer_amd/jpeg_decode.py). Run here, where the reference checkout is mounted and Pillow (libjpeg-turbo) is installed;
the tests themselves need neither.
  * the reference's two small JPEG assets as data (utopian/data/textures/defaults/checker.jpg: baseline 4:2:0, 225 x 225;
    prototype/data/models/FlightHelmet/screenshot/screenshot.jpg: progressive 4:4:4, 130 x 130) with the RGB Pillow decodes;
  * for each of the 65 JPEG textures of prototype/data/models/Sponza/glTF (baseline 4:4:4, 1024 x 1024; too large to commit):
    name, size, mean colour and CRC-32 of the RGB Pillow decodes - checked against the in-repo decoders where the checkout is mounted;
  * synthetic JPEGs written by Pillow from a seeded image: every chroma layout it offers x baseline / progressive / optimised
    tables x awkward sizes, grey, restart intervals, low and high quality, RGB-in-JFIF-less Adobe files, and a CMYK file (refused)."""
import glob
import io
import os
import zlib

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_fixtures.npz")


def pillow_rgb(data):
    im = Image.open(io.BytesIO(data))
    return np.asarray(im.convert("RGB") if im.mode != "L" else im, dtype=np.uint8)


def standard_tables():
    """the Annex K Huffman tables, read out of a file Pillow wrote without optimisation (libjpeg emits exactly those)"""
    buf = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8), "RGB").save(buf, format="JPEG", quality=75, subsampling="4:2:0")
    d, pos, tabs = buf.getvalue(), 2, {}
    while pos < len(d) and d[pos + 1] != 0xDA:
        L = (d[pos + 2] << 8) | d[pos + 3]
        if d[pos + 1] == 0xC4:
            at = pos + 4
            while at < pos + 2 + L:
                counts = list(d[at + 1:at + 17])
                n = sum(counts)
                tabs[d[at]] = (counts, list(d[at + 17:at + 17 + n]))
                at += 17 + n
        pos += 2 + L
    assert sorted(tabs) == [0x00, 0x01, 0x10, 0x11]
    return tabs


def encode_baseline(rgb, factors, tabs, interleaved=True, restart=0, qscale=1.0):
    """a plain baseline JPEG writer (JFIF, float DCT, Annex K tables) for layouts Pillow cannot write: `factors` = (h, v) per component"""
    import struct

    from scipy.fft import dctn

    zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    H, W = rgb.shape[:2]
    f = rgb.astype(np.float64)
    ycc = [0.299 * f[..., 0] + 0.587 * f[..., 1] + 0.114 * f[..., 2], 128 - 0.168736 * f[..., 0] - 0.331264 * f[..., 1] + 0.5 * f[..., 2],
           128 + 0.5 * f[..., 0] - 0.418688 * f[..., 1] - 0.081312 * f[..., 2]]
    hmax, vmax = max(h for h, _ in factors), max(v for _, v in factors)
    mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    qbase = [np.full(64, 1.0) * 6 + np.arange(64) * 0.9, np.full(64, 1.0) * 9 + np.arange(64) * 1.4]
    qts = [np.clip(np.round(q * qscale), 1, 65535).astype(int) for q in qbase]  # in zigzag order
    wide = any(q.max() > 255 for q in qts)
    codes = {}
    for key, (counts, vals) in tabs.items():
        code, k, m = 0, 0, {}
        for length in range(1, 17):
            for _ in range(counts[length - 1]):
                m[vals[k]] = (code, length)
                code += 1
                k += 1
            code <<= 1
        codes[key] = m
    planes = []
    for ci, (h, v) in enumerate(factors):
        cw, ch = -(-W * h // hmax), -(-H * v // vmax)
        ys = np.minimum((np.arange(ch) * vmax) // v, H - 1)
        xs = np.minimum((np.arange(cw) * hmax) // h, W - 1)
        p = ycc[ci][ys][:, xs]  # point sampling: any down-sampler gives a valid test stream
        p = np.pad(p, ((0, my * v * 8 - ch), (0, mx * h * 8 - cw)), mode="edge")
        blocks = p.reshape(my * v, 8, mx * h, 8).transpose(0, 2, 1, 3) - 128.0
        coef = dctn(blocks, type=2, norm="ortho", axes=(2, 3)).reshape(my * v, mx * h, 64)
        q = qts[0 if ci == 0 else 1]
        natural = np.zeros(64)
        natural[zz] = q
        planes.append((np.round(coef / natural).astype(int)[..., zz], cw, ch))  # zigzag order
    bits = []

    def put(code, length):
        bits.append((code, length))

    def flush_bits():
        acc, n, outb = 0, 0, bytearray()
        for code, length in bits:
            acc = (acc << length) | code
            n += length
            while n >= 8:
                n -= 8
                byte = (acc >> n) & 0xFF
                outb.append(byte)
                if byte == 0xFF:
                    outb.append(0)
        if n:
            byte = ((acc << (8 - n)) | ((1 << (8 - n)) - 1)) & 0xFF
            outb.append(byte)
            if byte == 0xFF:
                outb.append(0)
        bits.clear()
        return bytes(outb)

    def encode_block(b, ci, pred):
        dc, ac = codes[0x00 if ci == 0 else 0x01], codes[0x10 if ci == 0 else 0x11]
        diff = int(b[0]) - pred
        s = abs(diff).bit_length()
        put(*dc[s])
        if s:
            put(diff if diff > 0 else diff + (1 << s) - 1, s)
        run = 0
        last = max([k for k in range(1, 64) if b[k] != 0], default=0)
        for k in range(1, last + 1):
            v = int(b[k])
            if v == 0:
                run += 1
                continue
            while run > 15:
                put(*ac[0xF0])
                run -= 16
            s = abs(v).bit_length()
            put(*ac[(run << 4) | s])
            put(v if v > 0 else v + (1 << s) - 1, s)
            run = 0
        if last < 63:
            put(*ac[0x00])
        return int(b[0])

    def scan(comp_ids):
        header = bytes([len(comp_ids)]) + b"".join(bytes([c + 1, 0x00 if c == 0 else 0x11]) for c in comp_ids) + bytes([0, 63, 0])
        body = bytearray()
        pred = {c: 0 for c in comp_ids}
        count, rst = 0, 0
        if len(comp_ids) > 1:
            units = [(yy, xx) for yy in range(my) for xx in range(mx)]
        else:
            c = comp_ids[0]
            units = [(yy, xx) for yy in range(-(-planes[c][2] // 8)) for xx in range(-(-planes[c][1] // 8))]
        for (yy, xx) in units:
            if restart and count == restart:
                body += flush_bits() + bytes([0xFF, 0xD0 + rst])
                rst, count = (rst + 1) & 7, 0
                pred = {c: 0 for c in comp_ids}
            if len(comp_ids) > 1:
                for c in comp_ids:
                    h, v = factors[c]
                    for vv in range(v):
                        for hh in range(h):
                            pred[c] = encode_block(planes[c][0][yy * v + vv, xx * h + hh], c, pred[c])
            else:
                c = comp_ids[0]
                pred[c] = encode_block(planes[c][0][yy, xx], c, pred[c])
            count += 1
        body += flush_bits()
        return b"\xff\xda" + struct.pack(">H", 2 + len(header)) + header + bytes(body)

    out = bytearray(b"\xff\xd8\xff\xe0" + struct.pack(">H", 16) + b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    for tq, q in enumerate(qts):
        body = bytes([(16 if wide else 0) | tq]) + (b"".join(struct.pack(">H", int(x)) for x in q) if wide else bytes(int(x) for x in q))
        out += b"\xff\xdb" + struct.pack(">H", 2 + len(body)) + body
    sof = bytes([8]) + struct.pack(">HH", H, W) + bytes([3]) + b"".join(bytes([c + 1, (h << 4) | v, 0 if c == 0 else 1]) for c, (h, v) in enumerate(factors))
    out += b"\xff\xc0" + struct.pack(">H", 2 + len(sof)) + sof
    for key, (counts, vals) in sorted(tabs.items()):
        body = bytes([key]) + bytes(counts) + bytes(vals)
        out += b"\xff\xc4" + struct.pack(">H", 2 + len(body)) + body
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)
    if interleaved:
        out += scan([0, 1, 2])
    else:
        for c in range(3):
            out += scan([c])
    return bytes(out + b"\xff\xd9")


def main():
    out = {}
    names = []
    for key, rel in (("ref_checker", "utopian/data/textures/defaults/checker.jpg"), ("ref_screenshot", "prototype/data/models/FlightHelmet/screenshot/screenshot.jpg")):
        data = open(os.path.join(REF, rel), "rb").read()
        out[key + "_jpg"] = np.frombuffer(data, dtype=np.uint8)
        out[key + "_rgb"] = pillow_rgb(data)
        names.append(key)
    sponza = []
    for f in sorted(glob.glob(os.path.join(REF, "prototype/data/models/Sponza/glTF/*.jpg"))):
        rgb = pillow_rgb(open(f, "rb").read())
        sponza.append((os.path.basename(f), rgb.shape[1], rgb.shape[0], float(rgb[..., 0].mean()), float(rgb[..., 1].mean()), float(rgb[..., 2].mean()), zlib.crc32(rgb.tobytes())))
    out["sponza_names"] = np.array([s[0] for s in sponza])
    out["sponza_stats"] = np.array([s[1:6] for s in sponza], dtype=np.float64)
    out["sponza_crc"] = np.array([s[6] for s in sponza], dtype=np.uint64)

    rng = np.random.default_rng(20261004)

    def picture(w, h):
        y, x = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 120 * np.sin(x / 5.0 + y / 9.0), 127 + 120 * np.cos(x / 3.0 - y / 7.0), (x * 7 + y * 13) % 256], axis=2)
        base[h // 3:h // 3 + 3, :, :] = 255  # hard edges
        base[:, w // 2:w // 2 + 2, :] = 0
        return np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)

    def save(key, img, mode="RGB", **kw):
        buf = io.BytesIO()
        Image.fromarray(img, mode).save(buf, format="JPEG", **kw)
        data = buf.getvalue()
        out[key + "_jpg"] = np.frombuffer(data, dtype=np.uint8)
        out[key + "_rgb"] = pillow_rgb(data)
        names.append(key)

    sizes = [(1, 1), (7, 5), (8, 8), (17, 9), (33, 31), (64, 48), (100, 75)]
    for sub in ("4:4:4", "4:2:2", "4:2:0"):
        for si, (w, h) in enumerate(sizes):
            img = picture(w, h)
            tag = sub.replace(":", "")
            try:
                save(f"syn_{tag}_{w}x{h}_base", img, quality=88, subsampling=sub)
                if si % 2 == 0:
                    save(f"syn_{tag}_{w}x{h}_prog", img, quality=75, subsampling=sub, progressive=True)
                if si % 3 == 0:
                    save(f"syn_{tag}_{w}x{h}_opt", img, quality=93, subsampling=sub, optimize=True)
            except (ValueError, KeyError, OSError, TypeError) as e:
                print("skipped", sub, w, h, e)
    img = picture(96, 80)
    save("syn_grey", img[..., 0].copy(), mode="L", quality=85)
    save("syn_grey_prog", img[..., 1].copy(), mode="L", quality=60, progressive=True)
    save("syn_q10", img, quality=10, subsampling="4:2:0")
    save("syn_q100", img, quality=100, subsampling="4:4:4")
    save("syn_q3_prog", img, quality=3, subsampling="4:2:0", progressive=True)
    for blocks, tag in ((1, "rst1"), (5, "rst5")):
        try:
            save(f"syn_{tag}", img, quality=80, subsampling="4:2:0", restart_marker_blocks=blocks)
            save(f"syn_{tag}_prog", img, quality=80, subsampling="4:2:2", progressive=True, restart_marker_blocks=blocks)
        except TypeError as e:
            print("no restart markers in this Pillow:", e)
    try:
        save("syn_rst_rows", img, quality=80, subsampling="4:4:4", restart_marker_rows=1)
    except TypeError:
        pass
    # sampling layouts and scan structures Pillow's writer does not offer, from the small baseline encoder below; expected = Pillow's decode
    std = standard_tables()
    for tag, factors, kw in (("h1v2", [(1, 2), (1, 1), (1, 1)], {}), ("h4v1", [(4, 1), (1, 1), (1, 1)], {}), ("h2v2_noninterleaved", [(2, 2), (1, 1), (1, 1)], dict(interleaved=False)),
                             ("h1v1_noninterleaved_rst", [(1, 1), (1, 1), (1, 1)], dict(interleaved=False, restart=3)), ("h2v1_rst7", [(2, 1), (1, 1), (1, 1)], dict(restart=7)),
                             ("h4v2", [(4, 2), (1, 1), (1, 1)], {}), ("h2v4", [(2, 4), (1, 1), (1, 1)], {}), ("chroma_finer", [(1, 1), (2, 2), (2, 1)], {}), ("h3v1", [(3, 1), (1, 1), (1, 1)], {}),
                             ("h2v2_q16bit", [(2, 2), (1, 1), (1, 1)], dict(qscale=40.0))):
        for (w, h) in ((37, 29), (64, 64)):
            data = encode_baseline(picture(w, h), factors, std, **kw)
            try:
                rgb = pillow_rgb(data)
            except Exception as e:  # a layout libjpeg refuses is no fixture
                print("Pillow refuses", tag, e)
                continue
            key = f"enc_{tag}_{w}x{h}"
            out[key + "_jpg"], out[key + "_rgb"] = np.frombuffer(data, dtype=np.uint8), rgb
            names.append(key)
    buf = io.BytesIO()
    Image.fromarray(img, "RGB").convert("CMYK").save(buf, format="JPEG", quality=80)
    out["refuse_cmyk_jpg"] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
    out["names"] = np.array(names)
    np.savez_compressed(OUT, **out)
    print(len(names), "decodable fixtures,", len(sponza), "Sponza records ->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
