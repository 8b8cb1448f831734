"""CPU: the C++ glTF loader (include/utopian_gltf.hpp, what the C++ host mirror uses) against the Python loader
(rust-renderer_amd/gltf.py + image_decode.py, what the parity tests use) on glTF files written here - node hierarchies with
matrices and TRS, strided buffer views, 8/16/32-bit indices, normalised integer attributes, data URIs and external
files, PNG textures with every scanline filter, RGB / RGBA / palette - and, where the reference checkout is mounted, on
the reference's own assets. Neither side needs a GPU."""
import base64
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from rust_renderer_amd import gltf
from rust_renderer_amd.image_decode import UnsupportedImage, load_image_rgba8
from rust_renderer_amd.types import VERTEX_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/prototype/data/models"


@pytest.fixture(scope="module")
def dump_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("gltf") / "gltf_dump")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "gltf_dump.cpp"),
                    "-o", exe, "-lz"], check=True)
    return exe


def read_dump(path):
    raw = open(path, "rb").read()
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, raw, pos)
        pos += struct.calcsize("<" + fmt)
        return v

    n_meshes, n_tex = take("II")
    meshes, textures = [], []
    for _ in range(n_meshes):
        nv, ni, diffuse = take("III")
        base = take("4f")
        metallic, roughness = take("ff")
        transform = np.array(take("16f"), dtype=np.float32).reshape(4, 4).T  # column-major in the file
        (nl,) = take("I")
        name = raw[pos:pos + nl].decode()
        pos += nl
        v = np.frombuffer(raw, dtype=VERTEX_DTYPE, count=nv, offset=pos)
        pos += nv * 80
        idx = np.frombuffer(raw, dtype=np.uint32, count=ni, offset=pos)
        pos += ni * 4
        meshes.append(dict(vertices=v, indices=idx, diffuse=diffuse, base=base, metallic=metallic, roughness=roughness, transform=transform, name=name))
    for _ in range(n_tex):
        w, h = take("II")
        textures.append(np.frombuffer(raw, dtype=np.uint8, count=w * h * 4, offset=pos).reshape(h, w, 4))
        pos += w * h * 4
    assert pos == len(raw)
    return meshes, textures


def compare(dump_exe, path, tmp_path):
    out = str(tmp_path / "dump.bin")
    r = subprocess.run([dump_exe, path, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    meshes, textures = read_dump(out)
    py = gltf.load_gltf(path)
    assert len(meshes) == len(py.meshes) and len(textures) == len(py.textures)
    for c, p in zip(meshes, py.meshes):
        assert c["name"] == p.name
        assert np.array_equal(c["indices"], p.indices)
        assert c["vertices"].tobytes() == p.vertices.tobytes(), "vertex records differ"
        assert c["diffuse"] == (0xFFFFFFFF if p.texture is None else p.texture)
        assert np.array_equal(np.float32(c["base"]), np.float32(p.base_color))
        assert np.float32(c["metallic"]) == np.float32(p.metallic) and np.float32(c["roughness"]) == np.float32(p.roughness)
        # node products are taken in float32 on both sides, but not necessarily in the same order of additions
        assert np.allclose(c["transform"][:3, :].reshape(12), p.transform, rtol=0, atol=2e-6)
    for c, p in zip(textures, py.textures):
        assert np.array_equal(c, p)
    return meshes, textures


# ---- a PNG writer for the tests: chosen filter per row ---------------------------------------------------------------
def png_bytes(img, ctype, filters, palette=None, trns=None):
    h, w = img.shape[:2]
    rows = img.reshape(h, -1).astype(np.int32)
    bpp = rows.shape[1] // w
    out = bytearray()
    prev = np.zeros(rows.shape[1], dtype=np.int32)
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = rows[y]
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (a + prev) >> 1
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        out.append(ft)
        out += bytes(((cur - pred) & 255).astype(np.uint8))
        prev = cur

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette.astype(np.uint8).reshape(-1)))
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    comp = zlib.compress(bytes(out), 6)
    half = len(comp) // 2
    return data + chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")


def synthetic_gltf(tmp_path, rng):
    """two meshes under a small hierarchy; returns the .gltf path"""
    blob = bytearray()
    views, accessors = [], []

    def add(arr, ctype, kind, stride=0, normalized=False, pad=0):
        nonlocal blob
        while len(blob) % 4:
            blob.append(0)
        raw = arr.tobytes()
        item = arr.dtype.itemsize * (arr.shape[1] if arr.ndim == 2 else 1)
        if stride:
            rows = [raw[i * item:(i + 1) * item] + b"\xAB" * (stride - item) for i in range(len(arr))]
            raw = b"".join(rows)
        off = len(blob) + pad
        blob += b"\xCD" * pad + raw
        bv = {"buffer": 0, "byteOffset": off - pad, "byteLength": len(raw) + pad}
        if stride:
            bv["byteStride"] = stride
        views.append(bv)
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": len(arr), "type": kind}
        if pad:
            acc["byteOffset"] = pad
        if normalized:
            acc["normalized"] = True
        accessors.append(acc)
        return len(accessors) - 1

    def mesh(nv, ntri, index_dtype, index_ctype, with_uv, with_color, with_tangent, strided):
        pos = rng.normal(size=(nv, 3)).astype(np.float32)
        nrm = rng.normal(size=(nv, 3)).astype(np.float32)
        attrs = {"POSITION": add(pos, 5126, "VEC3", stride=20 if strided else 0), "NORMAL": add(nrm, 5126, "VEC3", pad=8 if strided else 0)}
        if with_uv:
            attrs["TEXCOORD_0"] = add(rng.integers(0, 65536, size=(nv, 2)).astype(np.uint16), 5123, "VEC2", normalized=True)
        if with_color:
            attrs["COLOR_0"] = add(rng.integers(0, 256, size=(nv, 3)).astype(np.uint8), 5121, "VEC3", stride=4, normalized=True)
        if with_tangent:
            attrs["TANGENT"] = add(rng.normal(size=(nv, 4)).astype(np.float32), 5126, "VEC4")
        idx = rng.integers(0, nv, size=ntri * 3).astype(index_dtype)
        return {"attributes": attrs, "indices": add(idx, index_ctype, "SCALAR")}

    p0 = mesh(37, 20, np.uint16, 5123, True, True, False, True)
    p0["material"] = 0
    p1 = mesh(12, 9, np.uint8, 5121, False, False, True, False)
    p1["material"] = 1
    p2 = mesh(300, 120, np.uint32, 5125, True, False, False, False)
    rgb = rng.integers(0, 256, size=(9, 7, 3)).astype(np.uint8)
    rgba = rng.integers(0, 256, size=(5, 6, 4)).astype(np.uint8)
    pal_idx = rng.integers(0, 5, size=(4, 8, 1)).astype(np.uint8)
    palette = rng.integers(0, 256, size=(5, 3))
    images = [
        {"uri": "data:image/png;base64," + base64.b64encode(png_bytes(rgb, 2, [0, 1, 2, 3, 4])).decode()},
        {"uri": "tex_rgba.png"},
        {"uri": "data:image/png;base64," + base64.b64encode(png_bytes(pal_idx, 3, [4, 1], palette=palette, trns=[10, 200, 255])).decode()},
    ]
    open(tmp_path / "tex_rgba.png", "wb").write(png_bytes(rgba, 6, [3, 4, 0]))
    q = np.array([0.1, -0.3, 0.2, 0.9], dtype=np.float64)
    q /= np.linalg.norm(q)
    doc = {
        "asset": {"version": "2.0"},
        "scenes": [{"nodes": [0, 3]}],
        "nodes": [
            {"name": "root é \"q\"", "children": [1, 2], "mesh": 1, "translation": [1.0, -2.0, 0.5], "scale": [2.0, 1.0, 0.5]},
            {"name": "child_trs", "mesh": 0, "rotation": [float(x) for x in q], "translation": [0.0, 3.0, 0.0]},
            {"name": "child_matrix", "mesh": 2, "matrix": [1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 4, 5, 6, 1]},
            {"name": "empty"},
        ],
        "meshes": [{"primitives": [p0, p1]}, {"primitives": [p1]}, {"primitives": [p2]}],
        "materials": [
            {"name": "textured", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.7, 0.6, 1.0], "baseColorTexture": {"index": 1}, "metallicFactor": 0.25, "roughnessFactor": 0.5}},
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 2}}},
        ],
        "textures": [{"source": 0}, {"source": 1}, {"source": 2}],
        "images": images,
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()}],
        "bufferViews": views,
        "accessors": accessors,
    }
    path = str(tmp_path / "synthetic.gltf")
    json.dump(doc, open(path, "w"), indent=1)
    return path


def test_cpp_loader_equals_python_loader_on_synthetic_files(dump_exe, tmp_path):
    rng = np.random.default_rng(7)
    path = synthetic_gltf(tmp_path, rng)
    meshes, textures = compare(dump_exe, path, tmp_path)
    # children before the node's own mesh; a primitive is named after its material, else after its node
    assert [m["name"] for m in meshes] == ["textured", "child_trs", "child_matrix", 'root \u00e9 "q"']
    assert len(textures) == 3 and textures[0].shape == (9, 7, 4) and (textures[0][..., 3] == 255).all() and textures[2].shape == (4, 8, 4)
    assert [m["diffuse"] for m in meshes] == [1, 2, 0xFFFFFFFF, 2]
    assert (meshes[1]["vertices"]["color"] == 1).all() and (meshes[0]["vertices"]["tangent"] == 0).all()
    assert np.float32(meshes[0]["metallic"]) == np.float32(0.25) and np.float32(meshes[0]["roughness"]) == np.float32(0.5)


def test_cpp_loader_reads_external_buffers(dump_exe, tmp_path):
    rng = np.random.default_rng(3)
    pos = rng.normal(size=(6, 3)).astype(np.float32)
    idx = np.arange(6, dtype=np.uint16)
    blob = pos.tobytes() + pos.tobytes() + idx.tobytes()
    open(tmp_path / "geo.bin", "wb").write(blob)
    doc = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1}, "indices": 2}]}],
           "buffers": [{"uri": "geo.bin", "byteLength": len(blob)}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 72}, {"buffer": 0, "byteOffset": 72, "byteLength": 72}, {"buffer": 0, "byteOffset": 144, "byteLength": 12}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 6, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 6, "type": "VEC3"},
                         {"bufferView": 2, "componentType": 5123, "count": 6, "type": "SCALAR"}]}
    path = str(tmp_path / "ext.gltf")
    json.dump(doc, open(path, "w"))
    meshes, _ = compare(dump_exe, path, tmp_path)
    assert np.array_equal(meshes[0]["vertices"]["pos"][:, :3], pos)


@pytest.mark.parametrize("ctype,channels", [(0, 1), (4, 2)])
def test_both_loaders_refuse_what_the_reference_refuses(dump_exe, tmp_path, ctype, channels):
    """grey and grey-alpha images: the reference's "Unsupported image format!" (gltf_loader.rs:196)"""
    img = np.random.default_rng(1).integers(0, 256, size=(3, 3, channels)).astype(np.uint8)
    data = png_bytes(img, ctype, [0])
    with pytest.raises(UnsupportedImage):
        load_image_rgba8(data)
    doc = {"asset": {"version": "2.0"}, "scenes": [{"nodes": []}], "images": [{"uri": "data:image/png;base64," + base64.b64encode(data).decode()}]}
    path = str(tmp_path / "grey.gltf")
    json.dump(doc, open(path, "w"))
    r = subprocess.run([dump_exe, path, str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "Unsupported image format" in r.stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "sphere.gltf")), reason="reference assets not mounted")
@pytest.mark.parametrize("name", ["sphere.gltf", "CornellBox-Original.gltf"])
def test_cpp_loader_equals_python_loader_on_the_reference_assets(dump_exe, tmp_path, name):
    meshes, _ = compare(dump_exe, os.path.join(REF, name), tmp_path)
    assert sum(len(m["indices"]) // 3 for m in meshes) == (4512 if name == "sphere.gltf" else 32)


def test_both_loaders_read_jpeg_textures(dump_exe, tmp_path):
    """JPEG images (65 of Sponza's 69): baseline 4:2:0 as a data URI, progressive as a file, a buffer-view image; a grey JPEG is
    the reference's "Unsupported image format!" like a grey PNG. Textures from the two loaders are compared byte for byte."""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_fixtures.npz"))
    base, prog, bview = (fx[k].tobytes() for k in ("ref_checker_jpg", "ref_screenshot_jpg", "syn_422_17x9_base_jpg"))
    open(tmp_path / "prog.jpg", "wb").write(prog)
    pos = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    blob = pos.tobytes() + np.uint16([0, 1, 2, 0]).tobytes() + bview
    doc = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 0}, "indices": 1, "material": 0}]}],
           "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 1}}}],
           "textures": [{"source": 0}, {"source": 1}, {"source": 2}],
           "images": [{"uri": "data:image/jpeg;base64," + base64.b64encode(base).decode()}, {"uri": "prog.jpg"}, {"bufferView": 2, "mimeType": "image/jpeg"}],
           "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}, {"buffer": 0, "byteOffset": 44, "byteLength": len(bview)}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}]}
    path = str(tmp_path / "jpeg.gltf")
    json.dump(doc, open(path, "w"))
    _, textures = compare(dump_exe, path, tmp_path)
    assert [t.shape for t in textures] == [(225, 225, 4), (130, 130, 4), (9, 17, 4)]
    assert np.array_equal(textures[0][..., :3], fx["ref_checker_rgb"]) and np.array_equal(textures[1][..., :3], fx["ref_screenshot_rgb"])
    assert all((t[..., 3] == 255).all() for t in textures)
    doc["images"] = [{"uri": "data:image/jpeg;base64," + base64.b64encode(fx["syn_grey_jpg"].tobytes()).decode()}]
    doc["textures"], doc["materials"] = [{"source": 0}], [{}]
    json.dump(doc, open(path, "w"))
    r = subprocess.run([dump_exe, path, str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "Unsupported image format" in r.stdout
    with pytest.raises(UnsupportedImage):
        gltf.load_gltf(path)


def test_cpp_loader_rejects_malformed_files_instead_of_reading_out_of_bounds(tmp_path):
    """short attribute accessors, indices beyond the vertices, cyclic node children, negative / non-finite offsets: an Error
    (the reference's loader panics), never an out-of-bounds read - the loader runs under ASan + UBSan here"""
    exe = str(tmp_path / "gltf_dump_asan")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "gltf_dump.cpp"), "-o", exe, "-lz"], check=True)
    pos = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]])
    blob = pos.tobytes() + pos[:2].tobytes() + np.uint16([0, 1, 2, 3, 2, 1]).tobytes()

    def doc(**changes):
        d = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
             "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 0}, "indices": 2}]}],
             "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
             "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 24}, {"buffer": 0, "byteOffset": 72, "byteLength": 12}],
             "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 2, "type": "VEC3"},
                           {"bufferView": 2, "componentType": 5123, "count": 6, "type": "SCALAR"}]}
        for k, v in changes.items():
            d[k] = v
        return d

    def run(d):
        path = str(tmp_path / "bad.gltf")
        json.dump(d, open(path, "w"))
        return subprocess.run([exe, path, str(tmp_path / "o.bin")], capture_output=True, text=True)

    assert run(doc()).returncode == 0
    short = doc()
    short["meshes"][0]["primitives"][0]["attributes"]["NORMAL"] = 1  # two normals for four positions
    r = run(short)
    assert r.returncode == 1 and "differ in element count" in r.stdout, r.stdout + r.stderr
    few = doc()
    few["accessors"][0]["count"] = 3  # index 3 now points past the vertices
    few["meshes"][0]["primitives"][0]["attributes"]["NORMAL"] = 0
    r = run(few)
    assert r.returncode == 1 and "index beyond" in r.stdout, r.stdout + r.stderr
    r = run(doc(nodes=[{"mesh": 0, "children": [1]}, {"children": [0]}]))
    assert r.returncode == 1 and "cyclic" in r.stdout, r.stdout + r.stderr
    neg = doc()
    neg["bufferViews"][0]["byteOffset"] = -8
    r = run(neg)
    assert r.returncode == 1 and "negative" in r.stdout, r.stdout + r.stderr
    huge = doc()
    huge["accessors"][0]["byteOffset"] = 1e300
    r = run(huge)
    assert r.returncode == 1, r.stdout + r.stderr
    # a byteStride whose product with the element count wraps size_t (2^52 x 4097 = 2^64 + 2^52) must not pass the bounds
    # check; the format allows strides of 4..252 only
    wrap = doc()
    wrap["bufferViews"][0]["byteStride"] = 2.0 ** 52
    wrap["accessors"][0]["count"] = 4097
    r = run(wrap)
    assert r.returncode == 1 and "byteStride" in r.stdout, r.stdout + r.stderr
    wide = doc()
    wide["bufferViews"][0]["byteStride"] = 16  # a legal stride, but four elements at 16 bytes need 60 of the buffer view's 48 + 36
    wide["accessors"][0]["count"] = 8
    r = run(wide)
    assert r.returncode == 1 and "past its buffer" in r.stdout, r.stdout + r.stderr
    ok16 = doc()
    ok16["bufferViews"][0]["byteStride"] = 16  # 3 x 16 + 12 = 60 <= 84 bytes of buffer: strided positions are read
    assert run(ok16).returncode == 0
