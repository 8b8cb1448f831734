"""The C++ host mirror (include/utopian_host.hpp) drives the same C ABI as the ctypes binding.
CPU: it compiles, links against libutopian_hip.so and fails loudly without a device.
GPU: frames rendered through Renderer::add_model / build_path_tracing_render_graph / Application
equal the ctypes path bit for bit."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import rust_renderer_amd as rr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_cpp(tmp_path):
    exe = str(tmp_path / "host_frames")
    libdir = os.path.dirname(rr.api.LIB_PATH)
    subprocess.run(
        ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_frames.cpp"),
         "-o", exe, "-L", libdir, "-lutopian_hip", f"-Wl,-rpath,{libdir}"],
        check=True,
    )
    return exe


def write_blob(path, scene, W, H, frames, pass_mask):
    view = scene.make_view(W, H)
    view.num_lights = len(scene.lights)
    cam = scene.camera
    with open(path, "wb") as f:
        f.write(struct.pack("<5I", 0x43534855, W, H, frames, pass_mask))
        f.write(bytes(view))
        f.write(struct.pack("<9f", *cam.position, *cam.target, cam.fov_degrees, cam.z_near, cam.z_far))
        textures, meshes = [], []
        for model, transform in scene.models:
            base = len(textures)
            textures += model.textures
            for m in model.meshes:
                w = m.transform if transform is None else rr.api.compose3x4(transform, m.transform)
                meshes.append((m, base, w))
        f.write(struct.pack("<I", len(textures)))
        for t in textures:
            f.write(struct.pack("<2I", t.shape[1], t.shape[0]))
            f.write(np.ascontiguousarray(t, dtype=np.uint8).tobytes())
        f.write(struct.pack("<I", len(meshes)))
        for m, base, w in meshes:
            f.write(struct.pack("<2I", len(m.vertices), len(m.indices)))
            f.write(np.ascontiguousarray(m.vertices).tobytes())
            f.write(np.ascontiguousarray(m.indices, dtype=np.uint32).tobytes())
            f.write(struct.pack("<i4fIf", -1 if m.texture is None else base + m.texture, *m.base_color, int(m.material_type), float(m.material_property)))
            mat4 = np.vstack([np.asarray(w, dtype=np.float32).reshape(3, 4), [0, 0, 0, 1]]).astype(np.float32)
            f.write(np.ascontiguousarray(mat4.T).tobytes())  # column-major
        f.write(struct.pack("<I", len(scene.lights)))
        for p in scene.lights:
            f.write(struct.pack("<3f", *p))
    return view


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_cpp_host_builds_links_and_fails_loudly_without_a_device(tmp_path):
    exe = build_cpp(tmp_path)
    scene = rr.scenes.cornell_scene(1, 4)
    blob = str(tmp_path / "scene.blob")
    write_blob(blob, scene, 32, 32, 1, rr.PASS_ALL)
    r = subprocess.run([exe, blob, str(tmp_path / "out.f32")], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr
    assert "status=2" in r.stdout and "no CPU fallback" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("pass_mask", [rr.PASS_ALL, rr.PASS_REFERENCE_PT])
def test_cpp_host_matches_ctypes_path(tmp_path, pass_mask):
    exe = build_cpp(tmp_path)
    W, H, frames = 96, 64, 3
    scene = rr.scenes.cornell_scene(2, 16)
    blob, out = str(tmp_path / "scene.blob"), str(tmp_path / "out.f32")
    write_blob(blob, scene, W, H, frames, pass_mask)
    r = subprocess.run([exe, blob, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    fields = dict(kv.split("=") for kv in r.stdout.split()[1:])
    assert int(fields["total_samples"]) == frames
    assert int(fields["passes"]) == (6 if pass_mask == rr.PASS_ALL else 1)
    assert float(fields["camera_max_diff"]) < 2e-5, "C++ Camera disagrees with the numpy glam mirror"
    acc_cpp = np.fromfile(out, dtype=np.float32).reshape(H, W, 4)

    gpu = scene.upload(rr.Renderer(W, H))
    loop = rr.FrameLoop(gpu, scene.make_view(W, H))
    for _ in range(frames):
        loop.frame(pass_mask)
    assert np.array_equal(acc_cpp.view(np.uint32), gpu.read_accumulation().view(np.uint32))
    assert int(fields["rays"]) == gpu.get_stats().path_rays
