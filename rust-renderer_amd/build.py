"""Builds libutopian_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU): every source to an object of its own
(in parallel, only the stale ones), then one link."""
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libutopian_hip.so")
OBJ_DIR = os.path.join(_HERE, "build")  # git-ignored and gpurun-ignored: only the .so travels

# -ffp-contract=off: the arithmetic contract fuses only where the source says fmaf().
HIPCC_FLAGS = [
    "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-function", "-pthread",
]


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_library(force=False, verbose=False, extra_flags=(), lib=LIB):
    """extra_flags: experiments (-D switches); such a build goes to another `lib` path and its own object directory"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(_HERE, "..", "include", "*.h")) + [os.path.abspath(__file__)]
    if not force and not _stale(lib, srcs + headers):
        return lib  # (also what a GPU box sees: the built .so travels with the snapshot, the objects do not)
    obj_dir = OBJ_DIR if lib == LIB else lib + ".obj"
    os.makedirs(obj_dir, exist_ok=True)
    flags = HIPCC_FLAGS + list(extra_flags) + ["-I", os.path.join(_HERE, "..", "include"), "-I", CSRC]
    objs = [os.path.join(obj_dir, os.path.basename(s) + ".o") for s in srcs]
    todo = [(s, o) for s, o in zip(srcs, objs) if force or _stale(o, [s] + headers)]

    def compile_one(so):
        cmd = [hipcc] + flags + ["-c", so[0], "-o", so[1]]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(compile_one, todo))
    if todo or not os.path.exists(lib):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-pthread", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return lib
