"""CPU: the two in-repo JPEG decoders of the glTF ingestion path (reference: gltf::import -> image crate,
utopian/src/gltf_loader.rs:168-196) - include/utopian_jpeg.hpp (C++ host mirror, built here with ASan + UBSan) and
rust-renderer_amd/jpeg_decode.py - against each other and against the committed fixtures of tests/golden/jpeg_fixtures.npz
(made by tests/golden/make_jpeg_fixtures.py: the reference's own small JPEG assets, synthetic files of every layout, and per-file
records of Sponza's 65 JPEG textures; expected pixels are libjpeg-turbo's). Both decoders restate the IJG integer arithmetic,
so the comparison is byte for byte, not "within 2 LSB"."""
import os
import subprocess
import zlib

import numpy as np
import pytest

from rust_renderer_amd.image_decode import UnsupportedImage, load_image_rgba8
from rust_renderer_amd.jpeg_decode import JpegError, decode_jpeg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPONZA = "/root/reference/prototype/data/models/Sponza/glTF"


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(ROOT, "tests", "golden", "jpeg_fixtures.npz"))


@pytest.fixture(scope="module")
def cpp(tmp_path_factory):
    d = tmp_path_factory.mktemp("jpeg")
    exe = str(d / "jpeg_dump")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "jpeg_dump.cpp"), "-o", exe], check=True)

    def run(blobs):
        """[bytes] -> [array or error string] through the C++ decoder"""
        lines = []
        for i, b in enumerate(blobs):
            open(d / f"in{i}.jpg", "wb").write(bytes(b))
            lines.append(f"{d / f'in{i}.jpg'} {d / f'out{i}.raw'}")
        (d / "list.txt").write_text("\n".join(lines) + "\n")
        r = subprocess.run([exe, str(d / "list.txt")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        out = []
        for i, line in enumerate(r.stdout.strip().split("\n")):
            if line.startswith("ok"):
                _, w, h, c, _ = line.split()
                out.append(np.fromfile(d / f"out{i}.raw", dtype=np.uint8).reshape(int(h), int(w), int(c)))
            else:
                out.append(line)
        assert len(out) == len(blobs)
        return out

    return run


def expected(fx, name):
    e = fx[name + "_rgb"]
    return e[:, :, None] if e.ndim == 2 else e


def test_fixtures_cover_the_format(fx):
    names = [str(n) for n in fx["names"]]
    assert len(names) >= 70
    info = {n: decode_jpeg(fx[n + "_jpg"].tobytes())[1] for n in names}
    layouts = {tuple(i["sampling"]) for i in info.values()}
    for need in ([(1, 1)] * 3, [(2, 1), (1, 1), (1, 1)], [(2, 2), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(4, 1), (1, 1), (1, 1)], [(1, 1)]):
        assert tuple(need) in layouts, need
    assert any(i["progressive"] for i in info.values()) and any(i["restart_interval"] for i in info.values())
    assert any(i["progressive"] and i["restart_interval"] for i in info.values())


def test_python_decoder_matches_libjpeg_byte_for_byte(fx):
    for n in (str(x) for x in fx["names"]):
        img, _ = decode_jpeg(fx[n + "_jpg"].tobytes())
        assert img.shape == expected(fx, n).shape, n
        assert np.array_equal(img, expected(fx, n)), (n, int(np.abs(img.astype(int) - expected(fx, n).astype(int)).max()))


def test_cpp_decoder_matches_libjpeg_and_python_byte_for_byte(fx, cpp):
    names = [str(x) for x in fx["names"]]
    for n, got in zip(names, cpp([fx[n + "_jpg"] for n in names])):
        assert not isinstance(got, str), (n, got)
        assert got.shape == expected(fx, n).shape and np.array_equal(got, expected(fx, n)), n


def test_the_reference_s_own_jpeg_assets(fx, cpp):
    """utopian/data/textures/defaults/checker.jpg (baseline 4:2:0, 225 x 225 - not a multiple of the 16 x 16 MCU) and
    prototype/data/models/FlightHelmet/screenshot/screenshot.jpg (progressive), committed as data"""
    for n, shape, prog in (("ref_checker", (225, 225, 3), False), ("ref_screenshot", (130, 130, 3), True)):
        img, info = decode_jpeg(fx[n + "_jpg"].tobytes())
        assert img.shape == shape and info["progressive"] == prog and np.array_equal(img, fx[n + "_rgb"])
        rgba = load_image_rgba8(fx[n + "_jpg"].tobytes())  # the loader's policy on top: RGB8 -> RGBA8, alpha 255
        assert rgba.shape == shape[:2] + (4,) and np.array_equal(rgba[..., :3], img) and (rgba[..., 3] == 255).all()


def test_refusals_and_damage(fx, cpp):
    cmyk = fx["refuse_cmyk_jpg"].tobytes()
    with pytest.raises(JpegError, match="4-component"):
        decode_jpeg(cmyk)
    good = fx["syn_420_33x31_base_jpg"].tobytes()
    grey = fx["syn_grey_jpg"].tobytes()
    with pytest.raises(UnsupportedImage, match="Unsupported image format"):
        load_image_rgba8(grey)  # the image crate reports L8; the reference loader panics on it (gltf_loader.rs:179-198)
    with pytest.raises(JpegError):
        decode_jpeg(b"\xff\xd8\xff\xd9")
    with pytest.raises(JpegError):
        decode_jpeg(good[:20])
    arithmetic = bytearray(good)
    at = arithmetic.index(b"\xff\xc0")
    arithmetic[at + 1] = 0xC9
    with pytest.raises(JpegError, match="arithmetic"):
        decode_jpeg(bytes(arithmetic))
    # damaged streams: every truncation and a set of byte flips go through the C++ decoder under ASan + UBSan - an error or an
    # image, never a crash; the Python twin gives the same verdict on the truncations (both feed zero bits past the end)
    rng = np.random.default_rng(9)
    cuts = [good[:k] for k in range(2, len(good), max(1, len(good) // 60))]
    flips = []
    for _ in range(60):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
        flips.append(bytes(b))
    got = cpp([cmyk, grey, bytes(arithmetic)] + cuts + flips)
    assert isinstance(got[0], str) and "4-component" in got[0]
    assert got[1].shape == (80, 96, 1)
    assert isinstance(got[2], str) and "arithmetic" in got[2]
    for blob, c in zip(cuts, got[3:3 + len(cuts)]):
        try:
            p, _ = decode_jpeg(blob)
        except (JpegError, IndexError):
            p = None
        if p is not None and not isinstance(c, str):
            assert np.array_equal(p, c)


@pytest.mark.skipif(not os.path.isdir(SPONZA), reason="the reference checkout is not mounted")
def test_sponza_textures_decode_to_the_recorded_images(fx, cpp):
    """all 65 JPEG textures of the reference's Sponza through the C++ decoder: size, mean colour and CRC-32 of the RGB image as
    recorded from libjpeg-turbo; three of them through the Python twin as well"""
    names = [str(n) for n in fx["sponza_names"]]
    assert len(names) == 65
    blobs = [open(os.path.join(SPONZA, n), "rb").read() for n in names]
    for i, (n, got) in enumerate(zip(names, cpp(blobs))):
        assert not isinstance(got, str), (n, got)
        w, h, r, g, b = fx["sponza_stats"][i]
        assert got.shape == (int(h), int(w), 3), n
        assert np.allclose(got.reshape(-1, 3).mean(0), [r, g, b], atol=1e-9), n
        assert zlib.crc32(got.tobytes()) == int(fx["sponza_crc"][i]), n
    for i in (0, 31, 64):
        img, info = decode_jpeg(blobs[i])
        assert zlib.crc32(img.tobytes()) == int(fx["sponza_crc"][i]) and info["sampling"] == [(1, 1)] * 3 and not info["progressive"]
