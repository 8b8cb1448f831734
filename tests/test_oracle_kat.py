"""CPU: the oracle against the known-answer vectors (tests/golden/rng_kat.json) and the analytic
identities derivable from the reference's shader text. The reference has no
tests of its own; see oracle/oracle.cpp's header for what this does and does not pin."""
import json
import os
import struct

import numpy as np
import pytest

import oracle_api as oa
import rust_renderer_amd as rr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
M = 0xFFFFFFFF


# -- a third, pure-Python restatement of random.glsl:5-34 (integers mod 2^32) ------------------
def py_jenkins(x):
    x = (x + (x << 10)) & M
    x ^= x >> 6
    x = (x + (x << 3)) & M
    x ^= x >> 11
    x = (x + (x << 15)) & M
    return x


def py_random_float(state):
    state = (state * 747796405 + 1) & M
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & M
    word = (word >> 22) ^ word
    f = struct.unpack("f", struct.pack("f", float(word)))[0] / 4294967296.0
    return struct.unpack("f", struct.pack("f", f))[0], state


def test_survey_hex_seeds():
    kat = json.load(open(os.path.join(GOLDEN, "rng_kat.json")))
    # the values SURVEY.md section 8c records in hex
    assert [hex(e["seed"]) for e in kat["init_rng"][:4]] == ["0xc0738807", "0xa5e9bdc", "0xc66e6241", "0x1b51ceb6"]
    assert {int(k): v for k, v in kat["jenkins_hash"].items()} == {0: 0, 1: 0x124EA49D, 2: 0x249DC93B, 0xDEADBEEF: 0x6C7328FE}


def test_rng_golden_vectors():
    kat = json.load(open(os.path.join(GOLDEN, "rng_kat.json")))
    for k, v in kat["jenkins_hash"].items():
        assert oa.jenkins_hash(int(k)) == v
    for e in kat["init_rng"]:
        seed = oa.init_rng(e["px"], e["py"], e["width"], e["frame"])
        assert seed == e["seed"]
        vals, state = oa.random_floats(seed, 3)
        assert np.allclose(vals, e["floats"], rtol=0, atol=5e-9)
        if "state_after" in e:
            assert state == e["state_after"]


def test_rng_three_restatements_agree():
    rng = np.random.default_rng(1)
    for x in rng.integers(0, 2**32, 200, dtype=np.uint64):
        x = int(x)
        assert oa.jenkins_hash(x) == py_jenkins(x) == int(rr.scenes.jenkins_hash(np.uint32(x)))
        vo, so = oa.random_floats(x, 4)
        s, vp = x, []
        for _ in range(4):
            v, s = py_random_float(s)
            vp.append(v)
        assert so == s and np.array_equal(vo, np.float32(vp))
        vn, sn = rr.scenes.pcg_float(np.uint32(x))
        assert float(vn) == float(vo[0])


def test_random_float_range_and_one_inclusive():
    # word = 0xffffffff converts to 2^32 in f32 -> exactly 1.0 ("between 0 and 1 inclusive", random.glsl:26)
    vals, _ = oa.random_floats(12345, 20000)
    assert vals.min() >= 0.0 and vals.max() <= 1.0
    assert abs(vals.mean() - 0.5) < 0.01


def test_random_point_in_unit_sphere_consumes_multiples_of_three():
    for seed in (1, 99, 0xDEADBEEF):
        p, s_after = oa.random_point_in_unit_sphere(seed)
        assert float(np.dot(p, p)) < 1.0
        s, draws = seed, 0
        while True:
            v = []
            for _ in range(3):
                f, s = py_random_float(s)
                v.append(np.float32(2.0) * np.float32(f) - np.float32(1.0))
            draws += 3
            if float(np.dot(np.float32(v), np.float32(v))) < 1.0:
                break
        assert s == s_after and draws % 3 == 0
        assert np.array_equal(p, np.float32(v))


def test_frame_number_is_total_samples_at_time_zero():
    v = rr.types.ViewUniformData()
    v.total_samples, v.time = 37, 0.0
    assert oa.frame_number(v) == 37
    v.time = 0.5
    assert oa.frame_number(v) == 5037  # int(float(total_samples) + time * 10000.0), reference.rgen:24


def test_seed_is_width_independent_for_origin_pixel():
    assert oa.init_rng(0, 0, 1920, 1) == oa.init_rng(0, 0, 256, 1)
    assert oa.init_rng(5, 3, 1920, 1) != oa.init_rng(5, 3, 256, 1)


def test_luminance_and_target_function_inverse_square():
    assert abs(oa.luminance((1, 1, 1)) - 1.0) < 2e-7  # 0.2126 + 0.7152 + 0.0722
    o = oa.OracleRenderer(4, 4)
    o.add_light((1.0, 2.0, 3.0))
    for p in ((0, 0, 0), (1, 2, 5), (-4, 0.5, 9)):
        d2 = sum((a - b) ** 2 for a, b in zip((1.0, 2.0, 3.0), p))
        assert abs(o.target_function(0, p) * d2 - 1.0) < 1e-5
    assert o.target_function(-1, (0, 0, 0)) == 0.0  # pinned: out-of-range light index has p_hat = 0
    assert o.target_function(1, (0, 0, 0)) == 0.0


def test_linear_to_srgb_identities():
    assert oa.linear_to_srgb(0.0) == 0.0
    assert abs(oa.linear_to_srgb(1.0) - 1.0) < 1e-6
    lo, hi = oa.linear_to_srgb(np.nextafter(np.float32(0.0031308), np.float32(0))), oa.linear_to_srgb(0.0031308)
    assert abs(lo - hi) < 1e-6  # continuity at the branch (view.glsl:52-60)
    xs = np.linspace(0, 1, 257, dtype=np.float32)
    ys = np.array([oa.linear_to_srgb(float(x)) for x in xs])
    assert (np.diff(ys) > 0).all()


def test_offset_ray_rt_gems_identities():
    n = np.float32([0.0, 1.0, 0.0])
    # |p| >= 1/32: integer offset of 256 * n ulps, away from the surface along n
    p = np.float32([2.0, 3.0, -4.0])
    q = oa.offset_ray(p, n)
    assert q[0] == p[0] and q[2] == p[2]
    assert q[1] > p[1] and (q[1].view(np.uint32) - p[1].view(np.uint32)) == 256
    # negative coordinate: the integer offset is subtracted so the point still moves along +n
    p = np.float32([2.0, -3.0, -4.0])
    q = oa.offset_ray(p, n)
    assert q[1] > p[1] and (p[1].view(np.uint32) - q[1].view(np.uint32)) == 256
    # |p| < 1/32: float offset n / 65536
    p = np.float32([0.01, 0.0, 0.02])
    q = oa.offset_ray(p, n)
    assert q[1] == np.float32(1.0 / 65536.0) and q[0] == p[0]
    # flipped normal flips the displacement
    q2 = oa.offset_ray(np.float32([2.0, 3.0, -4.0]), -n)
    assert q2[1] < 3.0


def test_sky_is_finite_positive_and_brighter_toward_the_sun():
    sun = (0.0, 0.9, 0.15)
    o = (0.0, 2.0, 0.0)
    up = oa.sky(o, (0.0, 1.0, 0.0), sun)
    horizon = oa.sky(o, (1.0, 0.02, 0.0), sun)
    away = oa.sky(o, (0.0, 0.3, -0.95), sun)
    for c in (up, horizon, away):
        assert np.isfinite(c).all() and (c >= 0).all()
    assert up[2] > up[0], "Rayleigh sky is blue overhead"
    toward = oa.sky(o, (0.0, 0.98, 0.17), sun)
    assert toward.sum() > away.sum()
    # mirror symmetry in x when the sun lies in the y-z plane
    a, b = oa.sky(o, (0.4, 0.5, 0.2), sun), oa.sky(o, (-0.4, 0.5, 0.2), sun)
    assert np.allclose(a, b, rtol=1e-5)


def test_sky_matches_the_independent_restatement_of_atmosphere_glsl():
    """the largest floating-point block of the path (atmosphere.glsl:53-214, ~400 exp per call) pinned by known answers: 48
    (origin, direction, sun) vectors - ground level, 1 km, inside the ozone layer, above the atmosphere (entry-point branch),
    toward / away from / below the sun, un-normalised bounce directions - evaluated by a numpy restatement written from the
    GLSL (tests/golden/make_sky_fixture.py), once with every operation in float32 (what the shader computes) and once in
    float64. The oracle matches the float32 evaluation to 3e-5 relative (libm vs numpy exp / pow in float32); the float64 one within the cancellation error the
    shader's own AtmosphereHeight carries (+-0.5 m on a 1,200 m scale height)."""
    import json
    import os

    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sky_kat.json")))
    assert len(kat["vectors"]) >= 32
    branches = set()
    for v in kat["vectors"]:
        sun = np.float32(v["sun"])
        sun = sun * (np.float32(1.0) / np.sqrt((sun[0] * sun[0] + sun[1] * sun[1]) + sun[2] * sun[2], dtype=np.float32))
        got = oa.sky(v["origin"], v["direction"], sun)  # IntegrateScattering itself; reference.rmiss:22 clamps it afterwards
        scale = max(float(np.max(np.abs(v["unclamped_f64"]))), 1e-9)
        assert np.abs(got - np.float64(v["unclamped_f32"])).max() <= 3e-5 * scale, v  # observed: <= 7e-6, 1.1e-5 looking into a sun on the horizon
        assert np.abs(got - np.float64(v["unclamped_f64"])).max() <= 5e-4 * scale, v
        assert np.array_equal(np.float64(v["sky_f64"]), np.minimum(np.float64(v["unclamped_f64"]), 1.0))
        branches.add((max(v["unclamped_f64"]) > 1.0, v["origin"][1] > 100000.0, max(v["sky_f64"]) < 1e-6))
    # the fixture reaches the clamp of reference.rmiss:22, the "ray starts above the atmosphere" branch and black sky (sun below)
    assert {b[0] for b in branches} == {True, False} and any(b[1] for b in branches)


def test_primary_ray_geometry():
    scene = rr.scenes.cornell_scene(1, 4)
    W, H = 64, 48
    v = scene.make_view(W, H)
    eye = np.float32([0.0, 0.9, 2.0])
    r = oa.primary_ray(v, W, H, W // 2, H // 2, 0.0, 0.0)  # jitter 0 at the centre pixel corner = NDC (0, 0)
    assert np.allclose(r[:3], eye, atol=1e-5)
    fwd = np.float32([0.0, 0.5, 0.0]) - eye
    fwd /= np.linalg.norm(fwd)
    assert np.allclose(r[3:], fwd, atol=1e-5)
    top = oa.primary_ray(v, W, H, W // 2, 0, 0.0, 0.0)
    assert top[4] > r[4], "row 0 is the top of the image (inUV.y = 1 - inUV.y, reference.rgen:33)"


def test_texture_sampler_mirrored_repeat_bilinear():
    o = oa.OracleRenderer(4, 4)
    tex = np.zeros((2, 2, 4), dtype=np.uint8)
    tex[0, 0] = (255, 0, 0, 255)
    tex[0, 1] = (0, 255, 0, 255)
    tex[1, 0] = (0, 0, 255, 255)
    tex[1, 1] = (255, 255, 255, 255)
    t = o.add_texture(tex)
    assert np.allclose(o.sample_texture(t, 0.25, 0.25), [1, 0, 0])  # texel centres
    assert np.allclose(o.sample_texture(t, 0.75, 0.25), [0, 1, 0])
    assert np.allclose(o.sample_texture(t, 0.5, 0.25), [0.5, 0.5, 0])  # halfway
    assert np.allclose(o.sample_texture(t, 0.0, 0.25), [1, 0, 0])  # edge: mirrored neighbour is the same texel
    assert np.allclose(o.sample_texture(t, 1.25, 0.25), o.sample_texture(t, 0.75, 0.25))  # mirror about u = 1
    assert np.allclose(o.sample_texture(t, -0.25, 0.75), o.sample_texture(t, 0.25, 0.75))  # mirror about u = 0


@pytest.mark.parametrize("mtype,prop", [(0, 0.0), (1, 0.3), (2, 1.5), (3, 0.0)])
def test_closest_hit_shader_material_contract(mtype, prop):
    """reference.rchit:46-89: per material, what is scattered, which colour is returned and how many
    random numbers are consumed from the payload seed."""
    o = oa.OracleRenderer(4, 4)
    verts, idx = rr.scenes.quad((-1, 0, -1), (0, 0, 2), (2, 0, 0), 1, 1)
    mat = rr.make_material(mtype, prop, (0.5, 0.25, 0.125, 1.0), o.default_diffuse_map())
    o.add_mesh(verts, idx, mat)
    o.initialize_raytracing()
    d = np.float32([0.3, -1.0, 0.2])
    seed = 777
    out, seed_after = o.closest_hit_shader(0, 0, 2.0, 0.25, 0.25, d, seed)
    color, t, scatter, scattered, normal = out[0:3], out[3], out[4:7], out[7], out[8:11]
    assert t == 2.0 and np.allclose(normal, [0, 1, 0], atol=1e-6), "normal faces the incoming ray"
    if mtype == 0:
        assert scattered == 1.0 and np.allclose(color, [0.5, 0.25, 0.125])
        p, s2 = oa.random_point_in_unit_sphere(seed)
        assert seed_after == s2 and np.array_equal(scatter, np.float32(normal + p))
    elif mtype == 1:
        assert scattered == 1.0 and np.allclose(color, 1.0)
        nd = d / np.linalg.norm(d)
        refl = nd - 2 * np.dot(nd, normal) * normal
        p, s2 = oa.random_point_in_unit_sphere(seed)
        assert seed_after == s2 and np.allclose(scatter, refl + prop * p, atol=1e-6)
    elif mtype == 2:
        assert scattered == 1.0 and np.allclose(color, 1.0)
        _, s2 = oa.random_floats(seed, 1)
        assert seed_after == s2, "dielectric draws exactly one number"
        assert abs(np.linalg.norm(scatter) - 1.0) < 1e-5
    else:
        assert scattered == 0.0 and np.allclose(color, 1.0) and seed_after == seed


def test_cook_torrance_extension_follows_the_reference_brdf():
    """material type 4 against a float64 evaluation of pbr_lighting.glsl:20-79 / brdf.glsl for the same N, V, L"""
    import ctypes as C
    import rust_renderer_amd as rr
    from rust_renderer_amd.scenes import Mesh, Model, Scene, _pack_vertices
    from rust_renderer_amd.camera import Camera
    pos = np.float32([[-1, -1, 0], [1, -1, 0], [0, 1, 0]])
    base, metallic, roughness = (0.8, 0.5, 0.3, 1.0), 0.35, 0.4
    m = Mesh(_pack_vertices(pos, np.tile(np.float32([0, 0, 1]), (3, 1)), np.zeros((3, 2), np.float32)), np.arange(3, dtype=np.uint32), rr.types.PBR, 0.0, base)
    m.metallic, m.roughness = metallic, roughness
    scene = Scene("one", [(Model([m], []), None)], [], Camera((0, 0, 4), (0, 0, 0), 60.0, 1.0, 0.01, 100.0))
    o = scene.upload(oa.OracleRenderer(8, 8))
    d = np.float32([0.3, -0.2, -1.0])
    seed0 = 12345
    out, _ = o.closest_hit_shader(0, 0, 1.5, 0.2, 0.3, d, seed0)
    # replay: same seed -> same point in the unit sphere -> L; then the reference formulas in float64
    p, _ = oa.random_point_in_unit_sphere(seed0)
    p = p.astype(np.float64)
    N = np.array([0, 0, 1.0]); V = -d.astype(np.float64) / np.linalg.norm(d); L = N + np.array(p); L /= np.linalg.norm(L)
    Hh = (V + L) / np.linalg.norm(V + L)
    a2 = roughness ** 4
    NDF = a2 / (np.pi * ((N @ Hh) ** 2 * (a2 - 1) + 1) ** 2)
    k = (roughness + 1) ** 2 / 8
    G = (N @ V) / ((N @ V) * (1 - k) + k) * (N @ L) / ((N @ L) * (1 - k) + k)
    b = np.array(base[:3]); F0 = 0.04 * (1 - metallic) + b * metallic
    F = F0 + (1 - F0) * (1 - max(Hh @ V, 0.0)) ** 5
    spec = NDF * G * F / (4 * (N @ V) * (N @ L) + 0.0001)
    want = (1 - F) * (1 - metallic) * b + spec * np.pi
    assert np.allclose(out[0:3], want, rtol=2e-5, atol=1e-6), (out, want)   # payload: color.rgb, distance, scatter.xyz, scattered, normal.xyz
    assert out[7] == 1.0
