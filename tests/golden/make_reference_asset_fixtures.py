"""Generates tests/golden/reference_assets.npz from the reference's own self-contained assets (run in the build
container, where /root/reference is mounted; the GPU box only ever sees the .npz):

  prototype/data/models/CornellBox-Original.gltf  (32 triangles, 8 meshes / materials, buffers embedded as base64)
  prototype/data/models/sphere.gltf               (4,512 triangles, embedded)
  utopian/data/textures/defaults/{white_texture,flat_normal_map,default_metallic_roughness}.png (Renderer::initialize)

through this repository's loader (rust-renderer_amd/gltf.py, image_decode.py). A fixture is data: per mesh the vertex array
(80-byte Vertex records), indices, base colour factor, node transform and name; the default textures as RGBA8.
The scene script that places them (prototype/src/scenes.rs:58-100) is restated in tests/test_reference_assets.py.

  python tests/golden/make_reference_asset_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rust_renderer_amd as rr  # noqa: E402
from rust_renderer_amd import gltf, image_decode  # noqa: E402

REF = "/root/reference"


def main():
    out = {}
    for key, path in (("cornell", "prototype/data/models/CornellBox-Original.gltf"), ("sphere", "prototype/data/models/sphere.gltf")):
        model = gltf.load_gltf(os.path.join(REF, path))
        assert not model.textures
        out[f"{key}_count"] = np.int32(len(model.meshes))
        for i, m in enumerate(model.meshes):
            out[f"{key}_{i}_vertices"] = m.vertices.view(np.uint8).reshape(-1, 80)
            out[f"{key}_{i}_indices"] = m.indices
            out[f"{key}_{i}_base_color"] = np.asarray(m.base_color, dtype=np.float32)
            out[f"{key}_{i}_transform"] = np.asarray(m.transform, dtype=np.float32)
            out[f"{key}_{i}_name"] = np.array(m.name)
    for name in ("white_texture", "flat_normal_map", "default_metallic_roughness"):
        img = image_decode.load_image_rgba8(open(os.path.join(REF, "utopian/data/textures/defaults", name + ".png"), "rb").read())
        # constant images (the normal map dithers by +-1 LSB): keep a 16x16 corner, enough to pin decode + RGB->RGBA expansion
        out[f"default_{name}"] = np.ascontiguousarray(img[:16, :16])
        out[f"default_{name}_shape"] = np.array(img.shape[:2], dtype=np.int32)
        out[f"default_{name}_unique"] = np.unique(img.reshape(-1, 4), axis=0)
    dst = os.path.join(ROOT, "tests", "golden", "reference_assets.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
