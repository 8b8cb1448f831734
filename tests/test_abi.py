"""CPU: the C-ABI shared library loads without a GPU, exports every entry point that
include/utopian_hip.h declares, refuses to run without a device (no CPU fallback), and its POD
structs have the byte layout of the reference structs."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

import rust_renderer_amd as rr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "utopian_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(uh_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("uh_create", "uh_destroy", "uh_add_texture_rgba8", "uh_add_mesh", "uh_add_light", "uh_set_instance_transform",
                 "uh_build_acceleration", "uh_render_frame", "uh_reset_accumulation", "uh_read_accumulation", "uh_read_output_bgra8",
                 "uh_read_reservoirs", "uh_get_stats", "uh_last_error", "uh_set_tile_partition", "uh_pack_tiles", "uh_unpack_tiles"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = rr.load_library()
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in utopian_hip.h but not exported: {missing}"
    assert b"gfx950" in lib.uh_version()
    assert hasattr(lib, "uh_rccl_gather_tiles") and hasattr(lib, "uh_hip_versions")


def test_library_links_hip_and_carries_gfx950_code():
    out = subprocess.run(["readelf", "-d", rr.api.LIB_PATH], capture_output=True, text=True).stdout
    assert "libamdhip64" in out
    blob = open(rr.api.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded"
    assert b"liboracle" not in blob and b"orc_render_frame" not in blob, "the product must not reference the oracle"


@pytest.mark.parametrize("which", [None, "system", "torch"])
def test_one_hip_runtime_per_process_and_it_is_the_one_the_library_was_built_with(which):
    """api._preload_hip_runtime: by DEFAULT (and with UH_HIP_RUNTIME=system) the library binds the HIP runtime of its RUNPATH -
    /opt/rocm's, the release whose hipcc compiled it, what a C / C++ / Rust host links - and it is the only one in the process;
    uh_hip_versions then reports the same major.minor twice. UH_HIP_RUNTIME=torch is the opt-in for sharing a process with PyTorch:
    the wheel's bundled copy is loaded first and the library binds to it - still ONE runtime, but of another release, and
    uh_version() says which (round 4's heap corruption under context churn showed in that configuration only)."""
    code = ("import sys, rust_renderer_amd as rr; lib = rr.load_library(); "
            "print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})); "
            "print(rr.hip_versions()); print(lib.uh_version().decode()); print('torch' in sys.modules)")
    env = {k: v for k, v in os.environ.items() if k != "UH_HIP_RUNTIME"}
    if which:
        env["UH_HIP_RUNTIME"] = which
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    paths, (built, runtime), version, torch_imported = eval(lines[-4]), eval(lines[-3]), lines[-2], eval(lines[-1])
    assert len(paths) == 1, f"two HIP runtimes in one process: {paths}"
    assert not torch_imported, "loading the library must not import torch"
    assert built in version and runtime in version
    import importlib.util
    has_torch_copy = importlib.util.find_spec("torch") is not None and os.path.exists(
        os.path.join(os.path.dirname(importlib.util.find_spec("torch").origin), "lib", "libamdhip64.so"))
    if which == "torch" and has_torch_copy:
        assert "/torch/lib/" in paths[0]
    else:
        assert "/torch/lib/" not in paths[0] and os.path.realpath(paths[0]).startswith(os.path.realpath("/opt/rocm") + "/"), paths
        assert built.split(".")[:2] == runtime.split(".")[:2], f"built with HIP {built}, running on {runtime}"


def test_versions_are_reported_without_a_gpu():
    lib = rr.load_library()
    b, r = C.c_int(0), C.c_int(0)
    assert lib.uh_hip_versions(C.byref(b), C.byref(r)) == 0 and lib.uh_hip_versions(None, None) == 0
    assert b.value // 10000000 >= 7 and r.value // 10000000 >= 7
    assert b"built with HIP" in lib.uh_version() and b"HIP runtime" in lib.uh_version()


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_device_fails_loudly():
    lib = rr.load_library()
    ctx = C.c_void_p()
    st = lib.uh_create(0, 64, 64, C.byref(ctx))
    assert st == 2 and not ctx.value  # UH_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.uh_last_error(None)
    with pytest.raises(rr.UtopianError, match="NO_DEVICE"):
        rr.Renderer(64, 64)


def test_null_context_is_rejected_not_dereferenced():
    lib = rr.load_library()
    api = rr.api.CApi(lib, "uh_")
    assert api.build_acceleration(None) == 1
    assert api.reset_accumulation(None) == 1
    assert api.render_frame(None, None, 0) == 1
    assert api.get_stats(None, None) == 1
    lib.uh_destroy(None)


def test_struct_layout_matches_header(tmp_path):
    """compile the header with gcc and compare sizeof / offsetof with the ctypes mirrors"""
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "utopian_hip.h"\n'
        "int main(void){\n"
        'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(UhVertex), sizeof(UhGpuMaterial), sizeof(UhGpuMesh), sizeof(UhGpuLight), sizeof(UhViewUniformData), sizeof(UhReservoir), sizeof(UhStats));\n'
        'printf("%zu %zu %zu %zu %zu\\n", offsetof(UhVertex, uv), offsetof(UhVertex, color), offsetof(UhVertex, tangent), offsetof(UhGpuMaterial, raytrace_properties), offsetof(UhGpuLight, intensity));\n'
        'printf("%zu %zu %zu %zu %zu %zu\\n", offsetof(UhViewUniformData, eye_pos), offsetof(UhViewUniformData, sun_dir), offsetof(UhViewUniformData, num_bounces), offsetof(UhViewUniformData, sky_enabled), offsetof(UhViewUniformData, temporal_reuse_enabled), offsetof(UhViewUniformData, raytracing_supported));\n'
        "return 0;}\n"
    )
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    sizes = [int(x) for x in lines[0].split()]
    T = rr.types
    assert sizes == [80, 64, 12, 96, 448, 16, C.sizeof(T.Stats)]
    assert sizes[:6] == [T.VERTEX_DTYPE.itemsize, C.sizeof(T.GpuMaterial), 12, C.sizeof(T.GpuLight), C.sizeof(T.ViewUniformData), C.sizeof(T.Reservoir)]
    assert [int(x) for x in lines[1].split()] == [32, 48, 64, 48, 64]
    assert [int(x) for x in lines[2].split()] == [320, 336, 352, 392, 412, 432]
    assert T.VERTEX_DTYPE.fields["uv"][1] == 32 and T.VERTEX_DTYPE.fields["tangent"][1] == 64


def test_header_carries_compile_time_layout_guards(tmp_path):
    """include/utopian_hip.h asserts every size / offset of SURVEY 8a A0 itself: it compiles as C11, as C89 (negative-array
    form) and as C++11, and a consumer whose packing differs (here: a 1-byte float, forced by a macro) does not compile"""
    src = tmp_path / "guard.c"
    src.write_text('#include "utopian_hip.h"\nint main(void){return 0;}\n')
    inc = ["-I", os.path.join(ROOT, "include")]
    for cmd in (["gcc", "-std=c11"], ["gcc", "-std=c89"], ["g++", "-std=c++11", "-x", "c++"]):
        subprocess.run(cmd + inc + ["-c", str(src), "-o", str(tmp_path / "guard.o")], check=True)
    text = open(HEADER).read()
    for name in ("UhVertex", "UhGpuMaterial", "UhGpuMesh", "UhGpuLight", "UhViewUniformData", "UhReservoir"):
        assert re.search(r"UH_LAYOUT_ASSERT\(sizeof\(%s\) == \d+" % name, text), name
    bad = subprocess.run(["gcc", "-std=c11", "-Dfloat=char"] + inc + ["-c", str(src), "-o", str(tmp_path / "bad.o")], capture_output=True, text=True)
    assert bad.returncode != 0 and "UhVertex" in bad.stderr
