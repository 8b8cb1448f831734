"""JPEG decoding for the Python host mirror's glTF loader - the twin of include/utopian_jpeg.hpp,
written from ITU-T T.81 and the published IJG algorithms. The reference loads textures through `gltf::import` -> the `image`
crate (utopian/src/gltf_loader.rs:168-196); 65 of Sponza's 69 images are baseline 8-bit 4:4:4 JPEGs.

Covers baseline (SOF0), extended sequential (SOF1) and progressive (SOF2) Huffman JPEGs, 8-bit, 1 or 3 components, interleaved and
non-interleaved scans, restart intervals, 8- / 16-bit quantisation tables, sampling factors 1..4. Entropy decoding is a plain
Python loop (seconds for a 1024^2 texture); everything after it is numpy and is the IJG reference decoder's integer arithmetic,
the same as the C++ twin's: "islow" inverse DCT, "fancy" triangle up-sampling for 2:1 chroma, fixed-point YCbCr -> RGB. The two
twins and libjpeg agree byte for byte (tests/test_jpeg.py)."""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)


class JpegError(ValueError):
    pass


def _huffman_lookup(counts, values):
    """16-bit look-ahead table: entry = (length << 8) | symbol, 0 where no code matches"""
    table = [0] * 65536
    code, k = 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            if code >= (1 << length):
                raise JpegError("Huffman table assigns more codes than its lengths allow")
            first = code << (16 - length)
            entry = (length << 8) | values[k]
            table[first:first + (1 << (16 - length))] = [entry] * (1 << (16 - length))
            code += 1
            k += 1
        code <<= 1
    return table


class _Bits:
    """entropy-coded segment: 0xFF00 un-stuffed up front; at a marker the data ends and zero bits are supplied"""

    def __init__(self, data, start):
        self.data, self.pos = data, start
        self.acc, self.n = 0, 0
        self._load_segment()

    def _load_segment(self):
        """bytes up to the next marker (RSTn included), with the stuffing removed"""
        d, i = self.data, self.pos
        while True:
            j = d.find(b"\xff", i)
            if j < 0 or j + 1 >= len(d):
                j = len(d)
                break
            if d[j + 1] == 0x00:
                i = j + 2
                continue
            break
        self.seg = d[self.pos:j].replace(b"\xff\x00", b"\xff")
        self.seg_pos = 0
        self.marker_at = j
        self.acc, self.n = 0, 0

    def fill(self, need):
        seg, p = self.seg, self.seg_pos
        while self.n < need:
            byte = seg[p] if p < len(seg) else 0
            p += 1
            self.acc = ((self.acc << 8) | byte) & 0xFFFFFFFFFFFF
            self.n += 8
        self.seg_pos = p

    def get(self, count):
        if count == 0:
            return 0
        if self.n < count:
            self.fill(count)
        self.n -= count
        return (self.acc >> self.n) & ((1 << count) - 1)

    def decode(self, table):
        if self.n < 16:
            self.fill(16)
        e = table[(self.acc >> (self.n - 16)) & 0xFFFF]
        if not e:
            raise JpegError("bad Huffman code")
        self.n -= e >> 8
        return e & 0xFF

    def restart(self, expected):
        """byte-align at an RSTn marker and go on behind it"""
        d, j = self.data, self.marker_at
        if j + 1 < len(d) and 0xD0 <= d[j + 1] <= 0xD7:
            self.pos = j + 2
            self._load_segment()
        else:
            self.seg, self.seg_pos, self.acc, self.n = b"", 0, 0, 0  # no marker where one is due: zeros from here on


def _extend(v, s):
    return v - (1 << s) + 1 if v < (1 << (s - 1)) else v


class _Component:
    pass


def _idct_islow(coef, q):
    """IJG jidctint.c jpeg_idct_islow over all blocks at once: coef (N, 64) natural order, q (64,) -> (N, 8, 8) uint8"""
    CB, P1 = 13, 2
    F = dict(a=2446, b=3196, c=4433, d=6270, e=7373, f=9633, g=12299, h=15137, i=16069, j=16819, k=20995, m=25172)

    def descale(x, n):
        return (x + (1 << (n - 1))) >> n

    def pass_1d(x, shift, axis_first):
        # x: (N, 8, 8); transforms along axis 1 (axis_first) or 2
        g = (lambda r: x[:, r, :]) if axis_first else (lambda r: x[:, :, r])
        z2, z3 = g(2), g(6)
        z1 = (z2 + z3) * F["c"]
        tmp2 = z1 + z3 * (-F["h"])
        tmp3 = z1 + z2 * F["d"]
        z2, z3 = g(0), g(4)
        tmp0 = (z2 + z3) << CB
        tmp1 = (z2 - z3) << CB
        tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
        tmp0, tmp1, tmp2, tmp3 = g(7), g(5), g(3), g(1)
        z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
        z5 = (z3 + z4) * F["f"]
        tmp0, tmp1, tmp2, tmp3 = tmp0 * F["a"], tmp1 * F["j"], tmp2 * F["m"], tmp3 * F["g"]
        z1, z2, z3, z4 = z1 * -F["e"], z2 * -F["k"], z3 * -F["i"] + z5, z4 * -F["b"] + z5
        tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
        outs = [tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3]
        return np.stack([descale(o, shift) for o in outs], axis=1 if axis_first else 2)

    x = (coef.astype(np.int64) * q.astype(np.int64)[None, :]).reshape(-1, 8, 8)
    ws = pass_1d(x, CB - P1, True)          # columns: along the row index
    out = pass_1d(ws, CB + P1 + 3, False)   # rows
    return np.clip(out + 128, 0, 255).astype(np.uint8)


def _upsample(plane, comp, W, H, hmax, vmax):
    """IJG jdsample.c: component plane (rows x padded width) -> (H, W)"""
    dw, dh = comp.width, comp.height
    if hmax % comp.h or vmax % comp.v:
        raise JpegError("fractional sampling ratios are not supported")
    hr, vr = hmax // comp.h, vmax // comp.v
    src = plane[:dh, :dw].astype(np.int32)
    if hr == 1 and vr == 1:
        return src[:H, :W].astype(np.uint8)

    def rows_near_far():
        ys = np.arange(H)
        sy = ys >> 1
        far = np.clip(np.where(ys & 1, sy + 1, sy - 1), 0, dh - 1)
        return src[np.clip(sy, 0, dh - 1)], src[far], ys

    if hr == 2 and vr == 1 and dw > 2:  # h2v1_fancy_upsample
        rows = src[np.clip(np.arange(H), 0, dh - 1)]
        out = np.empty((H, 2 * dw), dtype=np.int32)
        out[:, 0] = rows[:, 0]
        out[:, 1] = (rows[:, 0] * 3 + rows[:, 1] + 2) >> 2
        v = rows[:, 1:-1] * 3
        out[:, 2:-2:2] = (v + rows[:, :-2] + 1) >> 2
        out[:, 3:-2:2] = (v + rows[:, 2:] + 2) >> 2
        out[:, -2] = (rows[:, -1] * 3 + rows[:, -2] + 1) >> 2
        out[:, -1] = rows[:, -1]
        return out[:, :W].astype(np.uint8)
    if hr == 2 and vr == 2 and dw > 2:  # h2v2_fancy_upsample
        near, far, _ = rows_near_far()
        col = near * 3 + far  # (H, dw)
        out = np.empty((H, 2 * dw), dtype=np.int32)
        out[:, 0] = (col[:, 0] * 4 + 8) >> 4
        out[:, 1] = (col[:, 0] * 3 + col[:, 1] + 7) >> 4
        out[:, 2:-2:2] = (col[:, 1:-1] * 3 + col[:, :-2] + 8) >> 4
        out[:, 3:-2:2] = (col[:, 1:-1] * 3 + col[:, 2:] + 7) >> 4
        out[:, -2] = (col[:, -1] * 3 + col[:, -2] + 8) >> 4
        out[:, -1] = (col[:, -1] * 4 + 7) >> 4
        return out[:, :W].astype(np.uint8)
    if hr == 1 and vr == 2:  # h1v2_fancy_upsample (libjpeg-turbo)
        near, far, ys = rows_near_far()
        bias = np.where(ys & 1, 2, 1)[:, None]
        return ((near * 3 + far + bias) >> 2)[:, :W].astype(np.uint8)
    ys = np.clip(np.arange(H) // vr, 0, dh - 1)
    xs = np.clip(np.arange(W) // hr, 0, dw - 1)
    return src[ys][:, xs].astype(np.uint8)


def decode_jpeg(data):
    """bytes -> ((H, W, C) uint8 with C = 1 (grey) or 3 (RGB), info dict)"""
    try:
        return _decode(bytes(data))
    except IndexError as e:  # a header field read past the end of a cut-off file
        raise JpegError("truncated") from e


def _decode(data):
    if len(data) < 4 or data[:2] != b"\xff\xd8":
        raise JpegError("no SOI marker")
    qt, dc_tab, ac_tab = {}, {}, {}
    comps = []
    W = H = 0
    hmax = vmax = 1
    restart_interval = 0
    progressive = have_frame = jfif = adobe = False
    adobe_transform = -1
    pos = 2

    def be16(at):
        if at + 2 > len(data):
            raise JpegError("truncated")
        return (data[at] << 8) | data[at + 1]

    def decode_scan(header, entropy_start):
        ns = data[header]
        if not 1 <= ns <= 4:
            raise JpegError("bad SOS")
        sc = []
        for i in range(ns):
            cid, tables = data[header + 1 + 2 * i], data[header + 2 + 2 * i]
            c = next((k for k in comps if k.id == cid), None)
            if c is None:
                raise JpegError("scan names an unknown component")
            c.dc_table, c.ac_table, c.last_dc = tables >> 4, tables & 15, 0
            sc.append(c)
        Ss, Se = data[header + 1 + 2 * ns], data[header + 2 + 2 * ns]
        Ah, Al = data[header + 3 + 2 * ns] >> 4, data[header + 3 + 2 * ns] & 15
        if progressive and (Ss > Se or Se > 63 or (Ss == 0 and Se != 0) or (Ss > 0 and ns != 1) or Al > 13):
            raise JpegError("bad progressive scan parameters")
        for c in sc:
            if (not progressive or Ss == 0) and not (progressive and Ah) and c.dc_table not in dc_tab:
                raise JpegError("missing DC Huffman table")
            if (not progressive or Ss > 0) and c.ac_table not in ac_tab:
                raise JpegError("missing AC Huffman table")
        br = _Bits(data, entropy_start)
        zz = [int(z) for z in ZIGZAG]
        state = dict(eobrun=0)
        interleaved = ns > 1
        mcus_x = (W + 8 * hmax - 1) // (8 * hmax) if interleaved else sc[0].blocks_w
        mcus_y = (H + 8 * vmax - 1) // (8 * vmax) if interleaved else sc[0].blocks_h
        p1, m1 = 1 << Al, -(1 << Al)

        def block(c, by, bx):
            b = c.coef[by * c.stride_blocks + bx]
            if not progressive:
                s = br.decode(dc_tab[c.dc_table])
                if s > 15:
                    raise JpegError("bad DC category")
                c.last_dc += _extend(br.get(s), s) if s else 0
                b[0] = c.last_dc
                ac = ac_tab[c.ac_table]
                k = 1
                while k < 64:
                    rs = br.decode(ac)
                    r, s = rs >> 4, rs & 15
                    if s:
                        k += r
                        if k > 63:
                            raise JpegError("AC run past the end of the block")
                        b[zz[k]] = _extend(br.get(s), s)
                        k += 1
                    elif r != 15:
                        break
                    else:
                        k += 16
            elif Ss == 0:
                if Ah == 0:
                    s = br.decode(dc_tab[c.dc_table])
                    if s > 15:
                        raise JpegError("bad DC category")
                    c.last_dc += _extend(br.get(s), s) if s else 0
                    b[0] = c.last_dc * (1 << Al)
                elif br.get(1):
                    b[0] |= 1 << Al
            elif Ah == 0:
                if state["eobrun"] > 0:
                    state["eobrun"] -= 1
                    return
                ac = ac_tab[c.ac_table]
                k = Ss
                while k <= Se:
                    rs = br.decode(ac)
                    r, s = rs >> 4, rs & 15
                    if s:
                        k += r
                        if k > 63:
                            raise JpegError("AC run past the end of the block")
                        b[zz[k]] = _extend(br.get(s), s) * (1 << Al)
                    elif r == 15:
                        k += 15
                    else:
                        state["eobrun"] = (1 << r) + (br.get(r) if r else 0) - 1
                        break
                    k += 1
            else:
                ac = ac_tab[c.ac_table]
                k = Ss

                def correct(idx):
                    if br.get(1) and (b[idx] & p1) == 0:
                        b[idx] += p1 if b[idx] >= 0 else m1

                if state["eobrun"] == 0:
                    while k <= Se:
                        rs = br.decode(ac)
                        r, s = rs >> 4, rs & 15
                        if s:
                            s = p1 if br.get(1) else m1
                        elif r != 15:
                            state["eobrun"] = (1 << r) + (br.get(r) if r else 0)
                            break
                        while k <= Se:
                            idx = zz[k]
                            if b[idx] != 0:
                                correct(idx)
                            else:
                                r -= 1
                                if r < 0:
                                    break
                            k += 1
                        if s and k <= Se:
                            b[zz[k]] = s
                        k += 1
                if state["eobrun"] > 0:
                    while k <= Se:
                        idx = zz[k]
                        if b[idx] != 0:
                            correct(idx)
                        k += 1
                    state["eobrun"] -= 1

        until_restart, next_rst = restart_interval, 0
        for my in range(mcus_y):
            for mx in range(mcus_x):
                if restart_interval and until_restart == 0:
                    br.restart(next_rst)
                    next_rst = (next_rst + 1) & 7
                    until_restart = restart_interval
                    state["eobrun"] = 0
                    for c in sc:
                        c.last_dc = 0
                if interleaved:
                    for c in sc:
                        for v in range(c.v):
                            for h in range(c.h):
                                block(c, my * c.v + v, mx * c.h + h)
                else:
                    block(sc[0], my, mx)
                if restart_interval:
                    until_restart -= 1
        # the next marker that is not RSTn
        j = br.marker_at
        while j + 1 < len(data) and (data[j] != 0xFF or data[j + 1] in (0x00, 0xFF) or 0xD0 <= data[j + 1] <= 0xD7):
            j += 1
        return j

    while True:
        if pos + 4 > len(data):
            if have_frame:
                break
            raise JpegError("truncated before the frame header")
        if data[pos] != 0xFF:
            pos += 1
            continue
        m = data[pos + 1]
        if m == 0xFF:
            pos += 1
            continue
        if m == 0xD9:
            break
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            pos += 2
            continue
        L = be16(pos + 2)
        if L < 2 or pos + 2 + L > len(data):
            raise JpegError("marker segment reaches past the end of the data")
        seg, seg_end = pos + 4, pos + 2 + L
        if m in (0xC0, 0xC1, 0xC2):
            if have_frame:
                raise JpegError("more than one frame")
            progressive = m == 0xC2
            if data[seg] != 8:
                raise JpegError("only 8-bit precision is supported")
            H, W, nc = be16(seg + 1), be16(seg + 3), data[seg + 5]
            if W == 0 or H == 0:
                raise JpegError("empty image (DNL is not supported)")
            if nc not in (1, 3):
                raise JpegError("4-component (CMYK / YCCK) images are not supported" if nc == 4 else "component count")
            for i in range(nc):
                c = _Component()
                c.id, c.h, c.v, c.tq = data[seg + 6 + 3 * i], data[seg + 7 + 3 * i] >> 4, data[seg + 7 + 3 * i] & 15, data[seg + 8 + 3 * i]
                if not (1 <= c.h <= 4 and 1 <= c.v <= 4 and c.tq <= 3):
                    raise JpegError("bad component parameters")
                comps.append(c)
            if nc == 1:
                comps[0].h = comps[0].v = 1
            hmax, vmax = max(c.h for c in comps), max(c.v for c in comps)
            mcus_x, mcus_y = (W + 8 * hmax - 1) // (8 * hmax), (H + 8 * vmax - 1) // (8 * vmax)
            for c in comps:
                c.width, c.height = (W * c.h + hmax - 1) // hmax, (H * c.v + vmax - 1) // vmax
                c.blocks_w, c.blocks_h = (c.width + 7) // 8, (c.height + 7) // 8
                c.stride_blocks, c.rows_blocks = mcus_x * c.h, mcus_y * c.v
                if c.stride_blocks * c.rows_blocks > (1 << 24):
                    raise JpegError("image too large")
                c.coef = [[0] * 64 for _ in range(c.stride_blocks * c.rows_blocks)]
                c.last_dc = 0
            have_frame = True
        elif m in (0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise JpegError("lossless, hierarchical and arithmetic-coded JPEGs are not supported")
        elif m == 0xC4:
            at = seg
            while at < seg_end:
                tc, th = data[at] >> 4, data[at] & 15
                counts = list(data[at + 1:at + 17])
                n = sum(counts)
                if tc > 1 or th > 3 or n > 256 or at + 17 + n > seg_end:
                    raise JpegError("bad DHT")
                (ac_tab if tc else dc_tab)[th] = _huffman_lookup(counts, data[at + 17:at + 17 + n])
                at += 17 + n
        elif m == 0xDB:
            at = seg
            while at < seg_end:
                pq, tq = data[at] >> 4, data[at] & 15
                if pq > 1 or tq > 3 or at + 1 + 64 * (pq + 1) > seg_end:
                    raise JpegError("bad DQT")
                t = np.zeros(64, dtype=np.int64)
                for i in range(64):
                    t[ZIGZAG[i]] = be16(at + 1 + 2 * i) if pq else data[at + 1 + i]
                qt[tq] = t
                at += 1 + 64 * (pq + 1)
        elif m == 0xDD:
            restart_interval = be16(seg)
        elif m == 0xE0:
            jfif = jfif or data[seg:seg + 5] == b"JFIF\x00"
        elif m == 0xEE:
            if L >= 14 and data[seg:seg + 5] == b"Adobe":
                adobe, adobe_transform = True, data[seg + 11]
        elif m == 0xDA:
            if not have_frame:
                raise JpegError("SOS before SOF")
            pos = decode_scan(seg, seg_end)
            continue
        pos = seg_end
    if not have_frame:
        raise JpegError("no frame")

    planes = []
    for c in comps:
        if c.tq not in qt:
            raise JpegError("missing quantisation table")
        blocks = _idct_islow(np.array(c.coef, dtype=np.int64), qt[c.tq])  # (N, 8, 8)
        plane = blocks.reshape(c.rows_blocks, c.stride_blocks, 8, 8).transpose(0, 2, 1, 3).reshape(c.rows_blocks * 8, c.stride_blocks * 8)
        planes.append(_upsample(plane, c, W, H, hmax, vmax))
    info = dict(progressive=progressive, sampling=[(c.h, c.v) for c in comps], restart_interval=restart_interval)
    if len(comps) == 1:
        return planes[0][:, :, None], info
    ycc = True
    if jfif:
        ycc = True
    elif adobe:
        ycc = adobe_transform != 0
    elif [c.id for c in comps] == [ord("R"), ord("G"), ord("B")]:
        ycc = False
    if not ycc:
        return np.stack(planes, axis=2), info
    # jdcolor.c: 16-bit fixed point
    x = np.arange(256, dtype=np.int64) - 128
    cr_r, cb_b = (91881 * x + 32768) >> 16, (116130 * x + 32768) >> 16
    cr_g, cb_g = -46802 * x, -22554 * x + 32768
    y, cb, cr = (p.astype(np.int64) for p in planes)
    rgb = np.stack([y + cr_r[cr], y + ((cb_g[cb] + cr_g[cr]) >> 16), y + cb_b[cb]], axis=2)
    return np.clip(rgb, 0, 255).astype(np.uint8), info
