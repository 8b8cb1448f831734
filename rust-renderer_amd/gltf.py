"""glTF 2.0 ingestion on the caller side of the boundary: turns a
.gltf file into the `scenes.Model` that `Renderer.add_model` uploads, with the semantics of the
reference's loader (utopian/src/gltf_loader.rs:47-218):

* nodes are walked depth first, children BEFORE the node's own mesh (gltf_loader.rs:57-63), the
  node transform is parent * local, one Mesh + one transform per primitive;
* Vertex{pos.w = 0, normal.w = 0, uv (0,0) if absent, color (1,1,1,1) if absent, tangent 0 if absent};
* material: base colour factor / metallic / roughness; diffuse_map = the glTF *texture* index used to
  index the model's *image* list (the reference's own quirk, gltf_loader.rs:103-107 vs :183-207);
  material_type Lambertian, property 0 (callers override, e.g. scenes.rs:116-121);
* images become RGBA8 (RGB8 is expanded, anything else is the reference's "Unsupported image format!",
  gltf_loader.rs:179-198); PNG is decoded in-repo (image_decode.py), JPEG needs Pillow (optional).
Buffers may be base64 data URIs or files next to the .gltf.
`instance_transform_3x4` applies the scale-rotation-translation round trip of
Raytracing::fill_instance_array (raytracing.rs:229-248), which drops shear.
"""
import base64
import json
import os

import numpy as np

from .image_decode import load_image_rgba8
from .scenes import Mesh, Model
from .types import LAMBERTIAN, VERTEX_DTYPE

f32 = np.float32
_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NUM = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _load_uri(uri, base_dir):
    if uri.startswith("data:"):
        return base64.b64decode(uri.split(",", 1)[1])
    with open(os.path.join(base_dir, uri), "rb") as f:
        return f.read()


def _accessor(gltf, buffers, index, as_float=False):
    acc = gltf["accessors"][index]
    if "sparse" in acc:
        raise NotImplementedError("sparse accessors")
    dt, n, count = np.dtype(_COMPONENT[acc["componentType"]]), _NUM[acc["type"]], acc["count"]
    if "bufferView" not in acc:
        out = np.zeros((count, n), dtype=dt)
    else:
        bv = gltf["bufferViews"][acc["bufferView"]]
        raw = buffers[bv["buffer"]]
        start = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or dt.itemsize * n
        out = np.ndarray((count, n), dtype=dt, buffer=raw, offset=start, strides=(stride, dt.itemsize)).copy()
    if as_float and dt != np.float32:
        scale = {np.dtype(np.uint8): 255.0, np.dtype(np.uint16): 65535.0, np.dtype(np.int8): 127.0, np.dtype(np.int16): 32767.0}.get(dt)
        out = out.astype(f32) / f32(scale) if acc.get("normalized") and scale else out.astype(f32)
    return out


def _quat_to_mat3(q):
    x, y, z, w = (f32(v) for v in q)
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz, yy, yz, zz, wx, wy, wz = x * x2, x * y2, x * z2, y * y2, y * z2, z * z2, w * x2, w * y2, w * z2
    return np.array([[1 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1 - (xx + yy)]], dtype=f32)


def _node_matrix(node):
    if "matrix" in node:
        return np.array(node["matrix"], dtype=f32).reshape(4, 4).T  # glTF stores column-major
    m = np.eye(4, dtype=f32)
    r = _quat_to_mat3(node.get("rotation", [0, 0, 0, 1]))
    s = np.array(node.get("scale", [1, 1, 1]), dtype=f32)
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = np.array(node.get("translation", [0, 0, 0]), dtype=f32)
    return m


def load_gltf(path):
    """utopian::gltf_loader::load_gltf -> Model (textures as (H, W, 4) uint8 arrays)."""
    base_dir = os.path.dirname(os.path.abspath(path))
    with open(path) as f:
        gltf = json.load(f)
    buffers = [_load_uri(b["uri"], base_dir) for b in gltf.get("buffers", [])]
    model = Model([], [])
    for image in gltf.get("images", []):
        data = _load_uri(image["uri"], base_dir) if "uri" in image else bytes(
            buffers[gltf["bufferViews"][image["bufferView"]]["buffer"]][
                gltf["bufferViews"][image["bufferView"]].get("byteOffset", 0):][: gltf["bufferViews"][image["bufferView"]]["byteLength"]]
        )
        model.textures.append(load_image_rgba8(bytes(data)))

    def load_node(index, parent):
        node = gltf["nodes"][index]
        transform = (parent @ _node_matrix(node)).astype(f32)
        for child in node.get("children", []):
            load_node(child, transform)
        if "mesh" not in node:
            return
        for prim in gltf["meshes"][node["mesh"]]["primitives"]:
            attrs = prim["attributes"]
            pos = _accessor(gltf, buffers, attrs["POSITION"], True)
            nrm = _accessor(gltf, buffers, attrs["NORMAL"], True)
            idx = _accessor(gltf, buffers, prim["indices"]).astype(np.uint32).reshape(-1)
            v = np.zeros(len(pos), dtype=VERTEX_DTYPE)
            v["pos"][:, :3] = pos
            v["normal"][:, :3] = nrm
            if "TEXCOORD_0" in attrs:
                v["uv"] = _accessor(gltf, buffers, attrs["TEXCOORD_0"], True)
            if "TANGENT" in attrs:
                v["tangent"] = _accessor(gltf, buffers, attrs["TANGENT"], True)
            v["color"] = 1.0
            if "COLOR_0" in attrs:
                c = _accessor(gltf, buffers, attrs["COLOR_0"], True)
                v["color"][:, : c.shape[1]] = c
            mat = gltf["materials"][prim["material"]] if "material" in prim else {}
            pbr = mat.get("pbrMetallicRoughness", {})
            tex = pbr.get("baseColorTexture", {}).get("index")
            mesh = Mesh(v, idx, LAMBERTIAN, 0.0, tuple(float(x) for x in pbr.get("baseColorFactor", [1, 1, 1, 1])), tex, transform[:3, :].reshape(12).copy(),
                        name=mat.get("name", node.get("name", "")), metallic=float(pbr.get("metallicFactor", 1.0)), roughness=float(pbr.get("roughnessFactor", 1.0)))
            model.meshes.append(mesh)

    scenes = gltf.get("scenes", [])
    for scene in scenes:
        for n in scene.get("nodes", []):
            load_node(n, np.eye(4, dtype=f32))
    return model


def load_cube():
    """ModelLoader::load_cube (utopian/src/model_loader.rs:65-156): unit cube, 24 vertices, the
    reference's own face / normal / uv assignment (its "Top" face carries the -y normal)."""
    faces = [  # (normal, four corners in the reference's vertex order), uv = (0,1) (1,1) (1,0) (0,0)
        ((0, 0, 1), [(-.5, -.5, .5), (.5, -.5, .5), (.5, .5, .5), (-.5, .5, .5)]),
        ((0, 0, -1), [(-.5, -.5, -.5), (.5, -.5, -.5), (.5, .5, -.5), (-.5, .5, -.5)]),
        ((0, -1, 0), [(-.5, -.5, -.5), (.5, -.5, -.5), (.5, -.5, .5), (-.5, -.5, .5)]),
        ((0, 1, 0), [(-.5, .5, -.5), (.5, .5, -.5), (.5, .5, .5), (-.5, .5, .5)]),
        ((-1, 0, 0), [(-.5, -.5, -.5), (-.5, .5, -.5), (-.5, .5, .5), (-.5, -.5, .5)]),
        ((1, 0, 0), [(.5, -.5, -.5), (.5, .5, -.5), (.5, .5, .5), (.5, -.5, .5)]),
    ]
    uvs = [(0, 1), (1, 1), (1, 0), (0, 0)]
    v = np.zeros(24, dtype=VERTEX_DTYPE)
    for fi, (n, corners) in enumerate(faces):
        for ci, p in enumerate(corners):
            k = fi * 4 + ci
            v["pos"][k, :3], v["normal"][k, :3], v["uv"][k], v["color"][k] = p, n, uvs[ci], 1.0
    winding = [(2, 0, 1, 0, 2, 3), (0, 2, 1, 2, 0, 3)]  # front/top/right vs back/bottom/left pattern of model_loader.rs:76-99
    idx = []
    for fi, pat in enumerate([0, 1, 0, 1, 1, 0]):
        idx += [fi * 4 + o for o in winding[pat]]
    return Model([Mesh(v, np.array(idx, dtype=np.uint32), LAMBERTIAN, 0.0, (1.0, 1.0, 1.0, 1.0), None, name="cube")], [])


def instance_transform_3x4(world4x4):
    """instance.transform * model.transforms[i] -> to_scale_rotation_translation -> recomposed 3x4
    (raytracing.rs:229-248): scale = column lengths (x negated for a negative determinant),
    rotation = normalised columns through a quaternion, shear is lost."""
    m = np.asarray(world4x4, dtype=f32).reshape(4, 4) if np.size(world4x4) == 16 else np.vstack([np.asarray(world4x4, dtype=f32).reshape(3, 4), [0, 0, 0, 1]]).astype(f32)
    a = m[:3, :3].astype(np.float64)
    det = np.linalg.det(a)
    s = np.linalg.norm(a, axis=0)
    if det < 0:
        s[0] = -s[0]
    r = a / np.where(s == 0, 1.0, s)[None, :]
    # Mat3 -> Quat -> Mat3 (orthonormalises what is left)
    t = np.trace(r)
    if t > 0:
        w = np.sqrt(1 + t) / 2
        q = np.array([(r[2, 1] - r[1, 2]) / (4 * w), (r[0, 2] - r[2, 0]) / (4 * w), (r[1, 0] - r[0, 1]) / (4 * w), w])
    else:
        i = int(np.argmax(np.diag(r)))
        j, k = (i + 1) % 3, (i + 2) % 3
        x = np.sqrt(max(1 + r[i, i] - r[j, j] - r[k, k], 0.0)) / 2
        q = np.zeros(4)
        q[i], q[j], q[k], q[3] = x, (r[j, i] + r[i, j]) / (4 * x), (r[k, i] + r[i, k]) / (4 * x), (r[k, j] - r[j, k]) / (4 * x)
    q /= np.linalg.norm(q)
    out = np.zeros((3, 4), dtype=f32)
    out[:, :3] = _quat_to_mat3(q) * s.astype(f32)[None, :]
    out[:, 3] = m[:3, 3]
    return out.reshape(12)
