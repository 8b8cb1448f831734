"""Deterministic synthetic scenes for the BASELINE.json configs.

The reference's own assets cannot be used for the headline runs (Sponza.bin is absent from the
reference checkout, Bistro is not part of it), so each config is concretised as a seeded
procedural scene of the same class. Every random choice is drawn from the reference's own hash /
PCG functions (utopian/shaders/include/random.glsl:5-34) keyed by the stated seed, so scenes are
reproducible bit for bit with no files.

Scene scripts mirrored: prototype/src/scenes.rs:3-30 (lights), :58-100 (Cornell), :102-150
(Sponza + metal / dielectric spheres); camera defaults prototype/src/main.rs:44-52.
"""
from dataclasses import dataclass, field

import numpy as np

from .api import identity3x4, make_material, transform3x4
from .camera import Camera
from .types import DIELECTRIC, DIFFUSE_LIGHT, LAMBERTIAN, METAL, PBR, VERTEX_DTYPE

f32 = np.float32
u32 = np.uint32


# ---------------------------------------------------------------------------------------------
# hash-based PRNG (vectorised restatement of random.glsl:5-34)
# ---------------------------------------------------------------------------------------------
def jenkins_hash(x):
    x = np.asarray(x, dtype=u32).copy()
    with np.errstate(over="ignore"):
        x += x << u32(10)
        x ^= x >> u32(6)
        x += x << u32(3)
        x ^= x >> u32(11)
        x += x << u32(15)
    return x


def pcg_float(state):
    """one randomFloat() step from `state` (array); returns (value, new_state)."""
    with np.errstate(over="ignore"):
        s = np.asarray(state, dtype=u32) * u32(747796405) + u32(1)
        word = ((s >> ((s >> u32(28)) + u32(4))) ^ s) * u32(277803737)
        word = (word >> u32(22)) ^ word
    return (word.astype(np.float64) / 4294967296.0).astype(f32), s


def hash_floats(seed, n, stream=0):
    """n floats in [0,1]: value i = randomFloat(jenkins(i ^ jenkins(seed + stream)))."""
    with np.errstate(over="ignore"):
        key = jenkins_hash(np.array([(seed + stream * 0x9E3779B9) & 0xFFFFFFFF], dtype=u32))[0]
        st = jenkins_hash(np.arange(n, dtype=u32) ^ key)
    return pcg_float(st)[0]


# ---------------------------------------------------------------------------------------------
# containers
# ---------------------------------------------------------------------------------------------
@dataclass
class Mesh:
    vertices: np.ndarray
    indices: np.ndarray
    material_type: int = LAMBERTIAN
    material_property: float = 0.0
    base_color: tuple = (1.0, 1.0, 1.0, 1.0)
    texture: int = None  # index into Model.textures, None = default white map
    transform: np.ndarray = field(default_factory=identity3x4)
    name: str = ""
    metallic: float = 1.0   # GpuMaterial.metallic_factor / roughness_factor (renderer.rs:20-36); read by material type PBR only
    roughness: float = 1.0

    def material_struct(self):
        return make_material(self.material_type, self.material_property, self.base_color, metallic=self.metallic, roughness=self.roughness)

    @property
    def num_triangles(self):
        return len(self.indices) // 3


@dataclass
class Model:
    meshes: list = field(default_factory=list)
    textures: list = field(default_factory=list)


@dataclass
class Scene:
    name: str
    models: list  # [(Model, transform3x4 | None)]
    lights: list  # [(x, y, z)]
    camera: Camera
    view_flags: dict = field(default_factory=dict)
    # (resolution, lo, hi): an iso-surface mesh the renderer extracts itself, as mesh 0 - a HIP renderer on the GPU
    # (uh_add_isosurface_mesh); the oracle backend with its restatement of the reference's marching cubes, unless the very
    # triangles a HIP renderer extracted are handed over (device_mesh: image parity needs the same triangulation)
    device_isosurface: tuple = None
    device_triangles: int = 0
    device_mesh: tuple = None  # (vertices, indices) read back from the last HIP upload

    @property
    def num_triangles(self):
        return self.device_triangles + sum(m.num_triangles for model, _ in self.models for m in model.meshes)

    @property
    def num_meshes(self):
        return (1 if self.device_isosurface else 0) + sum(len(model.meshes) for model, _ in self.models)

    def upload(self, renderer):
        """Renderer::add_model / add_light for every model and light, then Raytracing::initialize."""
        if self.device_isosurface:
            res, lo, hi = self.device_isosurface
            if renderer.backend == "hip":
                mesh, self.device_triangles = renderer.add_isosurface_mesh(res, lo, hi)
                self.device_mesh = renderer.read_mesh(mesh) if mesh is not None else None
            elif self.device_mesh is not None and not renderer.backend.startswith("hip"):
                v, idx = self.device_mesh
                renderer.add_mesh(v, idx, make_material(base_color=(0.8, 0.8, 0.8, 1.0), diffuse_map=renderer.default_diffuse_map()), None)
            else:
                _, self.device_triangles = renderer.add_isosurface_mesh(res, lo, hi)
        for model, transform in self.models:
            renderer.add_model(model, transform)
        for p in self.lights:
            renderer.add_light(p, (1.0, 1.0, 1.0), 1.0)
        renderer.initialize_raytracing()
        return renderer

    def make_view(self, width, height, **overrides):
        from .api import default_view

        self.camera.aspect_ratio = width / height
        v = default_view(self.camera, width, height, num_lights=len(self.lights))
        flags = dict(self.view_flags)
        flags.update(overrides)
        for k, val in flags.items():
            setattr(v, k, val)
        return v


# ---------------------------------------------------------------------------------------------
# geometry builders
# ---------------------------------------------------------------------------------------------
def _pack_vertices(pos, nrm, uv):
    v = np.zeros(len(pos), dtype=VERTEX_DTYPE)
    v["pos"][:, :3] = pos
    v["pos"][:, 3] = 1.0
    v["normal"][:, :3] = nrm
    v["uv"] = uv
    v["color"] = 1.0
    return v


def param_surface(fn, nu, nv, uv_scale=(1.0, 1.0), flip=False):
    """Tessellate p = fn(s, t), (s, t) in [0,1]^2, into an (nu x nv)-quad grid. Normals from
    central differences of the tessellated surface (so displaced surfaces shade consistently)."""
    nu, nv = max(1, int(nu)), max(1, int(nv))
    s = np.linspace(0.0, 1.0, nu + 1, dtype=np.float64)
    t = np.linspace(0.0, 1.0, nv + 1, dtype=np.float64)
    S, T = np.meshgrid(s, t, indexing="ij")
    P = np.asarray(fn(S, T), dtype=np.float64)  # (nu+1, nv+1, 3)
    eps = 1e-4
    dS = np.asarray(fn(np.clip(S + eps, 0, 1), T), dtype=np.float64) - np.asarray(fn(np.clip(S - eps, 0, 1), T), dtype=np.float64)
    dT = np.asarray(fn(S, np.clip(T + eps, 0, 1)), dtype=np.float64) - np.asarray(fn(S, np.clip(T - eps, 0, 1)), dtype=np.float64)
    N = np.cross(dS, dT)
    ln = np.linalg.norm(N, axis=-1, keepdims=True)
    N = np.where(ln > 1e-20, N / np.maximum(ln, 1e-20), np.array([0.0, 1.0, 0.0]))
    if flip:
        N = -N
    uv = np.stack([S * uv_scale[0], T * uv_scale[1]], axis=-1)
    i = np.arange(nu)[:, None] * (nv + 1) + np.arange(nv)[None, :]
    a, b, c, d = i, i + (nv + 1), i + (nv + 1) + 1, i + 1
    tri = np.stack([a, b, c, a, c, d], axis=-1) if not flip else np.stack([a, c, b, a, d, c], axis=-1)
    verts = _pack_vertices(P.reshape(-1, 3).astype(f32), N.reshape(-1, 3).astype(f32), uv.reshape(-1, 2).astype(f32))
    return verts, tri.reshape(-1).astype(u32)


def merge(parts):
    vs, is_, base = [], [], 0
    for v, i in parts:
        vs.append(v)
        is_.append(i + u32(base))
        base += len(v)
    return np.concatenate(vs), np.concatenate(is_).astype(u32)


def icosphere(subdivisions):
    """unit icosphere; 20 * 4^subdivisions triangles; normal = position; equirect uv."""
    t = (1.0 + 5.0**0.5) / 2.0
    v = np.array(
        [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]],
        dtype=np.float64,
    )
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]],
        dtype=np.int64,
    )
    for _ in range(subdivisions):
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        edges.sort(axis=1)
        uniq, inv = np.unique(edges, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid])
        n = len(f)
        m01, m12, m20 = base + inv[:n], base + inv[n : 2 * n], base + inv[2 * n :]
        f = np.concatenate(
            [np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1), np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)]
        )
    uv = np.stack([0.5 + np.arctan2(v[:, 2], v[:, 0]) / (2 * np.pi), 0.5 - np.arcsin(np.clip(v[:, 1], -1, 1)) / np.pi], axis=1)
    return _pack_vertices(v.astype(f32), v.astype(f32), uv.astype(f32)), f.reshape(-1).astype(u32)


def quad(origin, eu, ev, nu=1, nv=1, bump=None, uv_scale=(1.0, 1.0), flip=False):
    origin, eu, ev = (np.asarray(a, dtype=np.float64) for a in (origin, eu, ev))
    n = np.cross(eu, ev)
    n /= np.linalg.norm(n)

    def fn(S, T):
        P = origin + S[..., None] * eu + T[..., None] * ev
        if bump is not None:
            P = P + bump(S, T)[..., None] * n
        return P

    return param_surface(fn, nu, nv, uv_scale, flip)


def box(center, half, n=1, uv_scale=(1.0, 1.0)):
    c, h = np.asarray(center, dtype=np.float64), np.asarray(half, dtype=np.float64)
    parts = []
    for axis in range(3):
        for sign in (-1.0, 1.0):
            a, b = (axis + 1) % 3, (axis + 2) % 3
            eu, ev = np.zeros(3), np.zeros(3)
            eu[a], ev[b] = 2 * h[a], 2 * h[b]
            o = c.copy()
            o[axis] += sign * h[axis]
            o[a] -= h[a]
            o[b] -= h[b]
            parts.append(quad(o, eu, ev, n, n, uv_scale=uv_scale, flip=sign < 0))
    return merge(parts)


def cylinder(base, radius, height, nseg, nstack, profile=None, flute=0.0, nflutes=0):
    bx, by, bz = base

    def fn(S, T):
        ang = 2 * np.pi * S
        r = radius * (profile(T) if profile is not None else 1.0)
        if flute > 0:
            r = r * (1.0 - flute * (0.5 + 0.5 * np.cos(nflutes * ang)))
        return np.stack([bx + r * np.cos(ang), by + height * T, bz - r * np.sin(ang)], axis=-1)

    return param_surface(fn, nseg, nstack, uv_scale=(4.0, 4.0 * height / max(radius * 6.28, 1e-3)))


def arch_band(p0, p1, rise, depth, nseg, nwidth):
    """half-ellipse band from p0 to p1 (same y), extruded `depth` along the horizontal normal."""
    p0, p1 = np.asarray(p0, dtype=np.float64), np.asarray(p1, dtype=np.float64)
    mid, half = 0.5 * (p0 + p1), 0.5 * (p1 - p0)
    along = half / np.linalg.norm(half)
    side = np.cross(along, [0.0, 1.0, 0.0])

    def fn(S, T):
        ang = np.pi * S
        P = mid - np.cos(ang)[..., None] * half + (rise * np.sin(ang))[..., None] * np.array([0.0, 1.0, 0.0])
        return P + ((T - 0.5) * depth)[..., None] * side

    return param_surface(fn, nseg, nwidth, uv_scale=(3.0, 1.0))


# ---------------------------------------------------------------------------------------------
# procedural RGBA8 textures
# ---------------------------------------------------------------------------------------------
def procedural_texture(seed, k, size):
    """albedo map k of a material set: brick / checker / stripes / value-noise, tinted."""
    y, x = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    u, v = x / size, y / size
    tint = 0.35 + 0.6 * hash_floats(seed, 3, stream=1000 + k).astype(np.float64)
    noise = hash_floats(seed, size * size, stream=2000 + k).reshape(size, size).astype(np.float64)
    kind = k % 4
    if kind == 0:  # running-bond brick
        row = np.floor(v * 16)
        uu = u * 8 + 0.5 * (row % 2)
        mortar = ((uu % 1.0) < 0.06) | (((v * 16) % 1.0) < 0.1)
        base = np.where(mortar, 0.55, 0.85)
    elif kind == 1:  # checker
        base = np.where((np.floor(u * 8) + np.floor(v * 8)) % 2 == 0, 0.9, 0.6)
    elif kind == 2:  # stripes (curtain fabric)
        base = 0.65 + 0.3 * np.sin(u * 2 * np.pi * 12)
    else:  # coarse value noise (stone)
        c = 16
        coarse = hash_floats(seed, c * c, stream=3000 + k).reshape(c, c).astype(np.float64)
        base = 0.6 + 0.35 * coarse[(y * c // size), (x * c // size)]
    base = np.clip(base * (0.92 + 0.08 * noise), 0.0, 1.0)
    rgb = np.clip(base[..., None] * tint[None, None, :], 0.0, 1.0)
    out = np.empty((size, size, 4), dtype=np.uint8)
    out[..., :3] = np.round(rgb * 255.0).astype(np.uint8)
    out[..., 3] = 255
    return out


# ---------------------------------------------------------------------------------------------
# config 1 — RTIOW 3 spheres + ground + 1 point light
# ---------------------------------------------------------------------------------------------
def rtiow_scene(subdivisions=5):
    sv, si = icosphere(subdivisions)

    def sphere(center, radius, mtype, prop, color):
        return Mesh(sv, si, mtype, prop, color, None, transform3x4((radius,) * 3, center), name="sphere")

    model = Model(
        [
            sphere((0.0, -100.5, -1.0), 100.0, LAMBERTIAN, 0.0, (0.8, 0.8, 0.0, 1.0)),
            sphere((0.0, 0.0, -1.0), 0.5, LAMBERTIAN, 0.0, (0.1, 0.2, 0.5, 1.0)),
            sphere((-1.0, 0.0, -1.0), 0.5, DIELECTRIC, 1.5, (1.0, 1.0, 1.0, 1.0)),
            sphere((1.0, 0.0, -1.0), 0.5, METAL, 0.0, (0.8, 0.6, 0.2, 1.0)),
        ],
        [],
    )
    cam = Camera((0.0, 0.0, 1.0), (0.0, 0.0, -1.0), 60.0, 1.0, 0.01, 1000.0)
    flags = dict(sky_enabled=1, sun_shadow_enabled=1, lights_enabled=1, use_ris_light_sampling=0, num_bounces=5, samples_per_frame=1)
    return Scene("rtiow", [(model, None)], [(0.0, 3.5, 0.0)], cam, flags)


# ---------------------------------------------------------------------------------------------
# Cornell-class box with a DiffuseLight cube (scenes.rs:58-100), all four material types
# ---------------------------------------------------------------------------------------------
def cornell_scene(subdivisions=2, tex_size=16):
    meshes = []
    tex = [procedural_texture(0xC0C0, k, tex_size) for k in range(3)]
    wall = dict(n=2)
    meshes.append(Mesh(*quad((-1, 0, -1), (0, 0, 2), (2, 0, 0), 4, 4, uv_scale=(2, 2)), base_color=(0.73, 0.73, 0.73, 1), texture=1, name="floor"))
    meshes.append(Mesh(*quad((-1, 2, -1), (2, 0, 0), (0, 0, 2), 2, 2), base_color=(0.73, 0.73, 0.73, 1), name="ceiling"))
    meshes.append(Mesh(*quad((-1, 0, -1), (2, 0, 0), (0, 2, 0), 2, 2), base_color=(0.73, 0.73, 0.73, 1), texture=0, name="back"))
    meshes.append(Mesh(*quad((-1, 0, -1), (0, 2, 0), (0, 0, 2), 2, 2), base_color=(0.65, 0.05, 0.05, 1), name="left"))
    meshes.append(Mesh(*quad((1, 0, -1), (0, 0, 2), (0, 2, 0), 2, 2), base_color=(0.12, 0.45, 0.15, 1), name="right"))
    del wall
    meshes.append(Mesh(*box((0, 0, 0), (1, 1, 1)), DIFFUSE_LIGHT, 0.0, (1, 1, 1, 1), None, transform3x4((0.5, 0.05, 0.35), (0.0, 1.95, 0.0)), name="light"))
    sv, si = icosphere(subdivisions)
    meshes.append(Mesh(sv, si, METAL, 0.1, (1, 1, 1, 1), None, transform3x4((0.3,) * 3, (-0.45, 0.3, -0.3)), name="metal"))
    meshes.append(Mesh(sv, si, DIELECTRIC, 1.5, (1, 1, 1, 1), None, transform3x4((0.3,) * 3, (0.45, 0.3, 0.2)), name="glass"))
    c, s = np.cos(0.4), np.sin(0.4)
    rot = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=f32)
    meshes.append(Mesh(*box((0, 0, 0), (1, 1, 1), 2), LAMBERTIAN, 0.0, (0.73, 0.73, 0.73, 1), 2, transform3x4((0.25, 0.5, 0.25), (0.1, 0.5, -0.45), rot), name="tall"))
    cam = Camera((0.0, 0.9, 2.0), (0.0, 0.5, 0.0), 60.0, 1.0, 0.01, 1000.0)
    lights = [(0.0, 1.6, 0.0), (-0.6, 0.4, 0.6), (0.7, 1.2, -0.5)]
    flags = dict(sky_enabled=1, sun_shadow_enabled=1, lights_enabled=1, use_ris_light_sampling=0)
    return Scene("cornell", [(Model(meshes, tex), None)], lights, cam, flags)


# ---------------------------------------------------------------------------------------------
# configs 2/3 — "Sponza-class" procedural atrium (seed 0x53504F4E): ~262 k triangles, 103 meshes,
# 25 Lambertian materials with procedural albedo maps; reference camera (scenes.rs:107-110)
# ---------------------------------------------------------------------------------------------
SPONZA_SEED = 0x53504F4E
TESS = 1.327  # base tessellation factor: detail=1.0 gives 262,432 triangles (Sponza.gltf: 262,267)
LIGHTS_SEED = 0x4C495445


def sponza_class_scene(detail=1.0, tex_size=1024, with_spheres=False, num_lights=0, sphere_subdivisions=4, target_meshes=103, num_materials=25,
                       num_textures=25, material_mix=False, seed=None, scene_name="sponza_class", cook_torrance=False):
    """detail scales every tessellation factor linearly (triangles ~ detail^2). detail=1.0 gives
    the headline ~262 k-triangle scene; tests use detail ~0.1."""
    seed = SPONZA_SEED if seed is None else seed
    rnd = hash_floats(seed, 4096, stream=7).astype(np.float64)
    ri = [0]

    def r():
        ri[0] += 1
        return rnd[ri[0] - 1]

    def n(x):
        return max(1, int(round(x * detail * TESS)))

    parts = []  # (name, verts, indices)

    def bumps(freq_u, freq_v, amp, phase=0.0):
        return lambda S, T: amp * (np.sin(S * freq_u * 2 * np.pi + phase) * np.sin(T * freq_v * 2 * np.pi + 1.3 * phase))

    def tiles(nu_, nv_, amp):
        return lambda S, T: amp * (np.minimum((S * nu_) % 1.0, (T * nv_) % 1.0) < 0.04)

    L, Wd, Hh = 16.0, 8.0, 14.0  # half length (x), half width (z), height
    # floor: 4 strips
    for k in range(4):
        x0 = -L + k * (2 * L / 4)
        parts.append(("floor", *quad((x0, 0, -Wd), (0, 0, 2 * Wd), (2 * L / 4, 0, 0), n(56), n(28), tiles(16, 8, -0.01), uv_scale=(4, 2))))
    # outer walls: 2 long walls x 3 bands, 2 end walls x 2 bands
    for sgn in (-1, 1):
        for k in range(3):
            y0, y1 = k * Hh / 3, (k + 1) * Hh / 3
            if sgn < 0:
                parts.append(("wall", *quad((-L, y0, -Wd), (2 * L, 0, 0), (0, y1 - y0, 0), n(96), n(14), bumps(24, 3, 0.015, r() * 6), uv_scale=(8, 1.2))))
            else:
                parts.append(("wall", *quad((L, y0, Wd), (-2 * L, 0, 0), (0, y1 - y0, 0), n(96), n(14), bumps(24, 3, 0.015, r() * 6), uv_scale=(8, 1.2))))
        for k in range(2):
            y0, y1 = k * Hh / 2, (k + 1) * Hh / 2
            if sgn < 0:
                parts.append(("endwall", *quad((-L, y0, Wd), (0, 0, -2 * Wd), (0, y1 - y0, 0), n(48), n(20), bumps(10, 4, 0.04, r() * 6), uv_scale=(4, 2))))
            else:
                parts.append(("endwall", *quad((L, y0, -Wd), (0, 0, 2 * Wd), (0, y1 - y0, 0), n(48), n(20), bumps(10, 4, 0.04, r() * 6), uv_scale=(4, 2))))
    # colonnades: two rows (z = +-3.5), two storeys
    col_x = np.linspace(-13.5, 13.5, 10)
    zrow = 3.5
    for sgn in (-1, 1):
        for storey, (y0, hgt, rad) in enumerate(((0.0, 4.2, 0.38), (5.0, 3.4, 0.28))):
            for cx in col_x:
                prof = lambda T: 1.0 + 0.25 * np.exp(-((T - 0.03) / 0.04) ** 2) + 0.3 * np.exp(-((T - 0.97) / 0.04) ** 2) - 0.08 * T
                parts.append(("column", *cylinder((cx, y0, sgn * zrow), rad, hgt, n(36), n(14), prof, 0.08, 16)))
            # arches between neighbouring columns
            for a, b in zip(col_x[:-1], col_x[1:]):
                parts.append(("arch", *arch_band((a + rad, y0 + hgt, sgn * zrow), (b - rad, y0 + hgt, sgn * zrow), 0.75, 0.8, n(20), n(4))))
        # gallery floor + ceiling slabs between colonnade and outer wall
        z0, z1 = sgn * zrow, sgn * Wd
        parts.append(("gallery_floor", *quad((-L, 5.0, z0), (2 * L, 0, 0), (0, 0, z1 - z0), n(64), n(10), tiles(32, 5, 0.008), uv_scale=(8, 1), flip=sgn > 0)))
        parts.append(("gallery_under", *quad((-L, 4.95, z0), (0, 0, z1 - z0), (2 * L, 0, 0), n(10), n(64), uv_scale=(1, 8), flip=sgn > 0)))
        parts.append(("gallery_roof", *quad((-L, 10.0, z0), (0, 0, z1 - z0), (2 * L, 0, 0), n(8), n(48), bumps(2, 12, 0.03), uv_scale=(1, 8), flip=sgn > 0)))
        parts.append(("cornice", *quad((-L, 10.0, z0), (2 * L, 0, 0), (0, 1.2, 0), n(96), n(6), bumps(40, 1, 0.05), uv_scale=(8, 0.5), flip=sgn < 0)))
    # curtains hanging between upper columns on both sides
    for sgn in (-1, 1):
        for j in range(5):
            a, b = col_x[2 * j], col_x[2 * j + 1]
            ph = r() * 6.28
            wave = lambda S, T, ph=ph: 0.12 * np.sin(S * 2 * np.pi * 5 + ph) * (0.3 + T) + 0.05 * np.sin(T * 2 * np.pi * 2 + ph)
            parts.append(("curtain", *quad((a + 0.3, 8.3, sgn * (zrow - 0.15)), (b - a - 0.6, 0, 0), (0, -3.1, 0), n(44), n(44), wave, uv_scale=(2, 2))))
    # vases with plants along the centre line, reliefs on the end walls
    sv, si = icosphere(max(0, int(round(3 + np.log2(max(detail, 0.05))))))
    for j in range(6):
        cx = -12.5 + j * 5.0
        cz = (r() - 0.5) * 1.5
        vv = sv.copy()
        vv["pos"][:, :3] = sv["pos"][:, :3] * f32([0.45, 0.6, 0.45]) + f32([cx, 0.6, cz])
        parts.append(("vase", vv, si))
        parts.append(("plant", *cylinder((cx, 1.1, cz), 0.05, 1.1 + r() * 0.6, n(10), n(8), lambda T: 1.0 + 2.0 * T * (1 - T))))
    for sgn in (-1, 1):
        relief = lambda S, T: 0.18 * np.exp(-(((S - 0.5) / 0.22) ** 2 + ((T - 0.5) / 0.22) ** 2)) * (1.0 + 0.3 * np.sin(S * 40) * np.sin(T * 40))
        if sgn < 0:
            parts.append(("relief", *quad((-L + 0.05, 1.5, 1.5), (0, 0, -3.0), (0, 3.0, 0), n(56), n(56), relief)))
        else:
            parts.append(("relief", *quad((L - 0.05, 1.5, -1.5), (0, 0, 3.0), (0, 3.0, 0), n(56), n(56), relief)))
    # roof beams across the open atrium strip (sky visible between them)
    for j in range(4):
        cx = -12.0 + j * 8.0
        parts.append(("beam", *box((cx, 11.6, 0.0), (0.25, 0.3, Wd), n(3))))

    # group the parts into exactly 103 meshes (Sponza.gltf has 103 primitives): columns, arches and
    # curtains are merged in runs so that the count lands on 103
    groups = []
    by_name = {}
    for name, v, i in parts:
        by_name.setdefault(name, []).append((v, i))
    singles = [(name, p) for name, ps in by_name.items() for p in ps]
    # deterministic greedy: merge consecutive same-name parts until the total equals the target
    runs = [[s] for s in singles]
    while len(runs) > target_meshes:
        # merge the two smallest adjacent runs of the same name
        best, best_k = None, -1
        for k in range(len(runs) - 1):
            if runs[k][0][0] != runs[k + 1][0][0]:
                continue
            size = sum(len(p[1][1]) for p in runs[k]) + sum(len(p[1][1]) for p in runs[k + 1])
            if best is None or size < best:
                best, best_k = size, k
        if best_k < 0:
            break
        runs[best_k] = runs[best_k] + runs[best_k + 1]
        del runs[best_k + 1]
    for run in runs:
        v, i = merge([p[1] for p in run])
        groups.append((run[0][0], v, i))

    textures = [procedural_texture(seed, k, tex_size) for k in range(num_textures)]
    meshes = []
    for gi, (mname, v, i) in enumerate(groups):
        k = (gi * 7 + 3) % num_materials
        col = 0.75 + 0.25 * hash_floats(seed, 3, stream=5000 + k).astype(np.float64)
        mtype, prop = LAMBERTIAN, 0.0
        if material_mix:  # "full PBR" of config 4 = the reference's four RT material types
            pick = hash_floats(seed, 2, stream=9000 + k).astype(np.float64)
            if mname in ("vase", "column") and pick[0] < 0.35:
                mtype, prop = METAL, float(0.05 + 0.4 * pick[1])
            elif mname in ("vase", "curtain") and pick[0] > 0.8:
                mtype, prop = DIELECTRIC, 1.5
        mesh = Mesh(v, i, mtype, prop, (float(col[0]), float(col[1]), float(col[2]), 1.0), k % num_textures, identity3x4(), name=mname)
        if cook_torrance and mtype == LAMBERTIAN:
            # extension: the diffuse surfaces become Cook-Torrance with per-mesh metallic / roughness
            mr = hash_floats(seed, 2, stream=9500 + k).astype(np.float64)
            mesh.material_type, mesh.metallic, mesh.roughness = PBR, float(mr[0] < 0.25) * float(0.5 + 0.5 * mr[1]), float(0.15 + 0.8 * mr[1])
        meshes.append(mesh)
    models = [(Model(meshes, textures), None)]

    if with_spheres:  # scenes.rs:116-149
        sv2, si2 = icosphere(sphere_subdivisions)
        models.append((Model([Mesh(sv2, si2, METAL, 0.0, (1, 1, 1, 1), None, identity3x4(), "metal_sphere")], []), transform3x4((0.6,) * 3, (-3.0, 2.65, 0.7))))
        models.append((Model([Mesh(sv2, si2, DIELECTRIC, 1.5, (1, 1, 1, 1), None, identity3x4(), "dielectric_sphere")], []), transform3x4((0.6,) * 3, (-3.0, 0.65, 0.7))))

    lights = []
    if num_lights:
        g = int(np.ceil(np.sqrt(num_lights)))
        jit = hash_floats(LIGHTS_SEED, 3 * g * g, stream=1).reshape(g * g, 3).astype(np.float64)
        for k in range(num_lights):
            ix, iz = k // g, k % g
            x = -L + 1.0 + (ix + jit[k, 0]) * (2 * L - 2.0) / g
            z = -Wd + 0.8 + (iz + jit[k, 2]) * (2 * Wd - 1.6) / g
            y = 0.6 + jit[k, 1] * 8.5
            lights.append((float(f32(x)), float(f32(y)), float(f32(z))))

    cam = Camera((-10.28, 2.10, -0.18), (0.0, 0.5, 0.0), 60.0, 16.0 / 9.0, 0.01, 1000.0)
    flags = dict(sky_enabled=1, sun_shadow_enabled=1, lights_enabled=1 if num_lights else 0, use_ris_light_sampling=1 if num_lights else 0)
    return Scene(scene_name, models, lights, cam, flags)


def scene_for_config(config, **kw):
    """BASELINE.json configs[config] concretised."""
    if config == 0:
        return rtiow_scene(kw.pop("subdivisions", 5))
    if config == 1:
        return sponza_class_scene(num_lights=0, **kw)
    if config == 2:
        return sponza_class_scene(num_lights=1024, **kw)
    if config == 3:
        return bistro_class_scene(**kw)
    if config == 4:
        kw.pop("tex_size", None)
        kw.pop("detail", None)
        return isosurface_scene(**kw)
    raise ValueError(f"config {config} is not defined in BASELINE.json")


# ---------------------------------------------------------------------------------------------
# config 5 - isosurface of the reference's marching-cubes density field
# ---------------------------------------------------------------------------------------------
def isosurface_scene(resolution=512):
    """BASELINE.json configs[4]: the 512^3 isosurface of the reference's density field (torus above a
    box, marching_cubes.comp:83-103 placed in a 32-unit domain), one Lambertian mesh on a ground plane, sky + sun.
    The mesh is extracted by the renderer the scene is uploaded to (Scene.upload): on the GPU by uh_add_isosurface_mesh
    (table-driven marching cubes, ~0.75 M triangles at 512^3, milliseconds)."""
    ground = Mesh(*quad((-64, 4.99, -64), (0, 0, 160), (160, 0, 0), 32, 32, uv_scale=(8, 8)), base_color=(0.6, 0.6, 0.6, 1.0), name="ground")
    cam = Camera((27.0, 19.0, 33.0), (16.0, 14.0, 16.0), 60.0, 16.0 / 9.0, 0.01, 1000.0)
    flags = dict(sky_enabled=1, sun_shadow_enabled=1, lights_enabled=0, use_ris_light_sampling=0)
    sc = Scene("isosurface", [(Model([ground], []), None)], [], cam, flags)
    sc.device_isosurface = (resolution, 0.0, 32.0)
    return sc


BISTRO_SEED = 0x42495354


def bistro_class_scene(detail=3.27, tex_size=1024, num_lights=64, cook_torrance=False):
    """config 4 of BASELINE.json ("Bistro Exterior full PBR, 3840x2160"): the same generator at
    ~3.3x tessellation (~2.8 M triangles), 130 meshes / materials with Metal and Dielectric mixed
    in, point lights on (uniform sampling). Bistro itself is not part of the reference checkout."""
    sc = sponza_class_scene(detail=detail, tex_size=tex_size, with_spheres=True, num_lights=num_lights, sphere_subdivisions=5, target_meshes=130,
                            num_materials=130, num_textures=25, material_mix=True, seed=BISTRO_SEED, scene_name="bistro_class", cook_torrance=cook_torrance)
    sc.view_flags.update(dict(lights_enabled=1, use_ris_light_sampling=0))
    return sc
