"""Builds libutopian_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libutopian_hip.so")

# -ffp-contract=off: the arithmetic contract fuses only where the source says fmaf().
HIPCC_FLAGS = [
    "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-function", "-pthread",
]


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_library(force=False, verbose=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(_HERE, "..", "include", "*.h"))
    if not force and not _stale(LIB, deps):
        return LIB
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", os.path.join(_HERE, "..", "include"), "-I", CSRC, "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB
