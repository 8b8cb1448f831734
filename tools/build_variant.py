#!/usr/bin/env python3
"""An experiment's library: csrc/kernels.hip compiled with extra -D switches, linked with the default build's other objects.
  python tools/build_variant.py <name> -DUH_X=1 ...   ->  rust-renderer_amd/libuh_<name>.so   (UTOPIAN_HIP_LIB / tools/ab.sh take it)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_renderer_amd as rr  # noqa: E402

pkg = os.path.dirname(rr.__file__) if not rr.__file__.endswith("rust_renderer_amd.py") else os.path.join(ROOT, "rust-renderer_amd")
sys.path.insert(0, pkg)
import build as b  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
b.build_library()
obj = os.path.join(b.OBJ_DIR, "kernels_%s.o" % name)
subprocess.run(["hipcc"] + b.HIPCC_FLAGS + flags + ["-I", os.path.join(ROOT, "include"), "-I", b.CSRC, "-c", os.path.join(b.CSRC, "kernels.hip"), "-o", obj], check=True)
others = [os.path.join(b.OBJ_DIR, f) for f in os.listdir(b.OBJ_DIR) if f.endswith(".o") and not f.startswith("kernels")]
lib = os.path.join(pkg, "libuh_%s.so" % name)
subprocess.run(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-pthread", "-o", lib, obj] + others, check=True)
print(lib)
