#!/usr/bin/env python3
"""Makes tests/golden/shading_kat.npz: a SECOND READING of the closest-hit shader and of the ReSTIR chain, written in numpy from
the reference's GLSL text - not from oracle/oracle.cpp, which it exists to check (VERDICT r3 missing 5 / do-this 6b).

Read here (reference checkout, utopian/shaders/):
  include/random.glsl:5-46                 jenkinsHash, initRNG, stepRNG, randomFloat, randomPointInUnitSphere
  pathtrace_reference/reference.rchit:12-92   schlick_reflectance, main (all four material branches)
  include/restir_sampling.glsl:59-131      get_light_intensity, target_function, sample_light_uniform, finalize_resampling,
                                           updateReservoir, resample
  restir/initial_ris.rgen:19-39, temporal_reuse.rgen:35-119, spatial_reuse.rgen:23-73, reset_reservoirs.comp:24-45
  include/view.glsl:46-51                  luminance

Arithmetic: every operation below is one IEEE binary32 operation (numpy float32 scalars), in the order the GLSL expression
tree gives, never fused. Where GLSL leaves a built-in's evaluation open, this file follows the conventions DESIGN.md section 2
("Arithmetic contract" and "Pinned choices") states for the whole repository - they are inputs of this reading, not taken from the
oracle's code:
  dot(a, b) = (a.x b.x + a.y b.y) + a.z b.z;  length(v) = sqrt(dot(v, v));  distance(a, b) = length(a - b);
  normalize(v) = v * (1 / sqrt(dot(v, v)));  reflect(I, N) = I - N * (2 * dot(N, I));
  refract(I, N, eta): k = 1 - eta eta (1 - dot(N, I)^2); k < 0 ? 0 : I eta - N (eta dot(N, I) + sqrt(k));
  pow(x, 5.0) = ((x x)(x x)) x;  pow(d, 2.0) = d d;  mix / vec * scalar componentwise;
  float(uint) / 4294967295.0f: the divisor rounds to 2^32 in binary32;
  texture(in_gbuffer_position, px / size) through the LINEAR sampler = ((a + b) + (c + d)) * 0.25 of the 2 x 2 texels up-left,
  index -1 mirrored to 0;  light index -1 or out of range: p_hat = 0;  uvec2(negative float) = (uint)(int)trunc(x), then the
  clamp sends the wrapped value to size - 1;  temporal index one past the end: clamped to W H - 1.
The closest-hit cases use the default white diffuse map (colour exactly 1, so the sampler stays out of it) and instance
transforms whose inverse is exact in any method (identity, axis permutations with signs, power-of-two scales).

Run anywhere (no reference checkout needed: this file IS the reading):  python tests/golden/make_shading_fixture.py"""
import os

import numpy as np

f32, u32, i32 = np.float32, np.uint32, np.int32
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shading_kat.npz")
np.seterr(over="ignore", invalid="ignore", divide="ignore")


# ---- random.glsl ---------------------------------------------------------------------------------------------
def jenkins_hash(x):
    x = u32(x)
    x = u32(x + u32(x << u32(10)))
    x = u32(x ^ (x >> u32(6)))
    x = u32(x + u32(x << u32(3)))
    x = u32(x ^ (x >> u32(11)))
    x = u32(x + u32(x << u32(15)))
    return x


def init_rng(px, py, res_x, frame):
    # uint(dot(pixelCoords, uvec2(1, resolution.x))) ^ jenkinsHash(frameNumber): the dot of uvec2s is evaluated in float
    d = f32(f32(px) * f32(1.0) + f32(py) * f32(res_x))
    return jenkins_hash(u32(u32(int(d)) ^ jenkins_hash(frame)))


class Rng:
    def __init__(self, state):
        self.s = u32(state)

    def random_float(self):
        self.s = u32(self.s * u32(747796405) + u32(1))
        s = self.s
        word = u32(u32((s >> u32((s >> u32(28)) + u32(4))) ^ s) * u32(277803737))
        word = u32((word >> u32(22)) ^ word)
        return f32(f32(word) / f32(4294967295.0))

    def point_in_unit_sphere(self):
        while True:
            a, b, c = self.random_float(), self.random_float(), self.random_float()
            p = np.array([f32(2) * a - f32(1), f32(2) * b - f32(1), f32(2) * c - f32(1)], dtype=f32)
            if dot(p, p) < f32(1):
                return p


# ---- the conventions ---------------------------------------------------------------------------------------------
def dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def length(v):
    return f32(np.sqrt(dot(v, v)))


def normalize(v):
    inv = f32(f32(1) / f32(np.sqrt(dot(v, v))))
    return (v * inv).astype(f32)


def reflect(I, N):
    return (I - N * f32(f32(2) * dot(N, I))).astype(f32)


def refract(I, N, eta):
    dn = dot(N, I)
    k = f32(f32(1) - f32(f32(eta * eta) * f32(f32(1) - f32(dn * dn))))
    if k < 0:
        return np.zeros(3, dtype=f32)
    return (I * eta - N * f32(f32(eta * dn) + f32(np.sqrt(k)))).astype(f32)


# ---- reference.rchit ---------------------------------------------------------------------------------------------
def schlick_reflectance(cosine, ref_idx):  # rchit:12-18
    r0 = f32(f32(f32(1) - ref_idx) / f32(f32(1) + ref_idx))
    r0 = f32(r0 * r0)
    x = f32(f32(1) - cosine)
    x5 = f32(f32(f32(x * x) * f32(x * x)) * x)
    return f32(r0 + f32(f32(f32(1) - r0) * x5))


def closest_hit(n0, n1, n2, attribs, world_to_object, ray_dir, mtype, prop, base_color, seed):
    """rchit:20-92 with the white default map; returns (color, scatterDirection, isScattered, world_normal, randomSeed)"""
    rng = Rng(seed)
    b = np.array([f32(f32(f32(1) - attribs[0]) - attribs[1]), attribs[0], attribs[1]], dtype=f32)          # :30
    normal = ((n0 * b[0] + n1 * b[1]).astype(f32) + (n2 * b[2]).astype(f32)).astype(f32)                      # :31
    # :32 vec3(normal * gl_WorldToObjectEXT): a row vector times the 4x3 matrix - component j = dot(normal, column j)
    wn = np.array([dot(normal, world_to_object[:, j]) for j in range(3)], dtype=f32)
    world_normal = normalize(wn)
    if dot(world_normal, ray_dir) > f32(0):                                                                     # :35-37
        world_normal = (-world_normal).astype(f32)
    color = (np.ones(3, dtype=f32) * base_color).astype(f32)                                                    # :40-41 (white map)
    scattered = False
    if mtype == 0:                                                                                              # :47-50
        scatter = (world_normal + rng.point_in_unit_sphere()).astype(f32)
        scattered = bool(dot(ray_dir, world_normal) < f32(0))
    elif mtype == 1:                                                                                            # :52-59
        scatter = reflect(normalize(ray_dir), world_normal)
        scatter = (scatter + (rng.point_in_unit_sphere() * prop).astype(f32)).astype(f32)
        scattered = True
        color = np.ones(3, dtype=f32)
    elif mtype == 2:                                                                                            # :61-83
        nd = normalize(ray_dir)
        dnd = dot(nd, world_normal)
        outward = (-world_normal).astype(f32) if dnd > 0 else world_normal
        ratio = prop if dnd > 0 else f32(f32(1) / prop)
        cos_theta = f32(min(dot((nd * f32(-1)).astype(f32), outward), f32(1)))
        sin_theta = f32(np.sqrt(f32(f32(1) - f32(cos_theta * cos_theta))))
        cannot_refract = bool(f32(ratio * sin_theta) > f32(1))
        reflectance = schlick_reflectance(cos_theta, ratio)
        if cannot_refract or reflectance > rng.random_float():
            scatter = reflect(nd, outward)
        else:
            scatter = refract(nd, outward, ratio)
        scattered = True
        color = np.ones(3, dtype=f32)
    else:                                                                                                       # :85-89
        scatter = np.zeros(3, dtype=f32)  # uninitialised in the shader: the raygen never reads it for an unscattered path
        scattered = False
        color = np.ones(3, dtype=f32)
    return color, scatter, scattered, world_normal, rng.s


# ---- restir_sampling.glsl ------------------------------------------------------------------------------------------
class Lights:
    def __init__(self, pos, intensity, num_lights, max_used):
        self.pos, self.intensity, self.num_lights, self.max_used = pos.astype(f32), intensity.astype(f32), int(num_lights), int(max_used)

    def target_function(self, index, hit_position):  # :64-69 + get_light_intensity :59-62 + luminance (view.glsl:47-51)
        if index < 0 or index >= len(self.pos):
            return f32(0)
        d = length((self.pos[index] - hit_position).astype(f32))
        d2 = f32(d * d)
        inten = (self.intensity[index] / d2).astype(f32)
        return dot(inten, np.array([0.2126, 0.7152, 0.0722], dtype=f32))

    def sample_uniform(self, rng):  # :71-77
        n = min(self.num_lights, self.max_used)
        index = int(f32(rng.random_float() * f32(n)))
        return index, f32(f32(1) / f32(n))


def new_reservoir():
    return {"Y": -1, "W_sum": f32(0), "W_X": f32(0), "M": 0}


def finalize_resampling(r, p_hat):  # :79-82
    r["W_X"] = f32(0) if p_hat == f32(0) else f32(f32(f32(f32(1) / p_hat) * r["W_sum"]) / f32(r["M"]))


def update_reservoir(rng, r, Xi, w_i, M):  # :85-94
    r["W_sum"] = f32(r["W_sum"] + w_i)
    r["M"] = int(r["M"] + M)
    if f32(rng.random_float() * r["W_sum"]) < w_i:
        r["Y"] = int(Xi)


def resample(lights, rng, hit_position):  # :96-131
    r = new_reservoir()
    M = 32
    for _ in range(M):
        cand, p = lights.sample_uniform(rng)
        m_i = f32(f32(1) / f32(M))
        p_hat = lights.target_function(cand, hit_position)
        W_Xi = f32(f32(1) / p)
        w_i = f32(f32(m_i * p_hat) * W_Xi)
        update_reservoir(rng, r, cand, w_i, 1)
    r["M"] = 1
    if r["Y"] != -1:
        finalize_resampling(r, lights.target_function(r["Y"], hit_position))
    return r


# ---- the raygens ---------------------------------------------------------------------------------------------------
def gbuffer_fetch(g, px, py):
    x0, y0 = max(px - 1, 0), max(py - 1, 0)
    a, b, c, d = g[y0, x0, :3], g[y0, px, :3], g[py, x0, :3], g[py, px, :3]
    return (((a + b).astype(f32) + (c + d).astype(f32)).astype(f32) * f32(0.25)).astype(f32)


def initial_ris(lights, g, W, H, frame):  # initial_ris.rgen:19-39
    out = [[None] * W for _ in range(H)]
    for py in range(H):
        for px in range(W):
            rng = Rng(init_rng(px, py, W, frame))
            hit = gbuffer_fetch(g, px, py)
            nr = new_reservoir()
            r = resample(lights, rng, hit)
            update_reservoir(rng, nr, r["Y"], f32(r["W_sum"] * f32(r["M"])), r["M"])
            finalize_resampling(nr, lights.target_function(nr["Y"], hit))
            out[py][px] = nr
    return out


def mat4_vec4(m, v):
    """column-major mat4 * vec4, ((c0 x + c1 y) + c2 z) + c3 w per row"""
    return np.array([f32(f32(f32(m[r] * v[0]) + f32(m[4 + r] * v[1])) + f32(m[8 + r] * v[2])) + f32(m[12 + r] * v[3]) for r in range(4)], dtype=f32)


def temporal_reuse(lights, g, W, H, frame, initial, prev, prev_pv, enabled=True):  # temporal_reuse.rgen:35-119
    out = [[None] * W for _ in range(H)]
    for py in range(H):
        for px in range(W):
            ir = initial[py][px]
            if not enabled:
                out[py][px] = dict(ir)
                continue
            rng = Rng(init_rng(px, py, W, frame))
            hit = gbuffer_fetch(g, px, py)
            nr = new_reservoir()
            p_hat = lights.target_function(ir["Y"], hit)
            update_reservoir(rng, nr, ir["Y"], f32(f32(p_hat * ir["W_X"]) * f32(ir["M"])), ir["M"])
            pr = new_reservoir()
            uvw = mat4_vec4(prev_pv, np.array([hit[0], hit[1], hit[2], f32(1)], dtype=f32))
            ux, uy = f32(uvw[0] / uvw[3]), f32(uvw[1] / uvw[3])
            ux, uy = f32(f32(ux * f32(0.5)) + f32(0.5)), f32(f32(uy * f32(0.5)) + f32(0.5))
            uy = f32(f32(1) - uy)
            if ux >= 0 and ux <= 1 and uy >= 0 and uy <= 1:
                ix, iy = int(f32(f32(ux * f32(W)) + f32(0.5))), int(f32(f32(uy * f32(H)) + f32(0.5)))
                ti = min(iy * W + ix, W * H - 1)
                pr = dict(prev[ti // W][ti % W])
            p_hat = f32(0) if pr["Y"] == -1 else lights.target_function(pr["Y"], hit)
            pr["M"] = min(20 * ir["M"], pr["M"])
            update_reservoir(rng, nr, pr["Y"], f32(f32(p_hat * pr["W_X"]) * f32(pr["M"])), pr["M"])
            if nr["Y"] != -1:
                finalize_resampling(nr, lights.target_function(nr["Y"], hit))
            out[py][px] = nr
    return out


def spatial_reuse(lights, g, W, H, frame, temporal, enabled=True):  # spatial_reuse.rgen:23-73
    out = [[None] * W for _ in range(H)]
    for py in range(H):
        for px in range(W):
            tr = temporal[py][px]
            if not enabled:
                out[py][px] = dict(tr)
                continue
            rng = Rng(init_rng(px, py, W, frame))
            hit = gbuffer_fetch(g, px, py)
            nr = new_reservoir()
            p_hat = lights.target_function(tr["Y"], hit)
            update_reservoir(rng, nr, tr["Y"], f32(f32(p_hat * tr["W_X"]) * f32(tr["M"])), tr["M"])
            for _ in range(5):
                ox = f32(f32(rng.random_float() * f32(2)) - f32(1))
                oy = f32(f32(rng.random_float() * f32(2)) - f32(1))
                ox, oy = f32(ox * f32(30)), f32(oy * f32(30))
                nx = (px + (int(ox) & 0xFFFFFFFF)) & 0xFFFFFFFF  # uvec2(offset): (uint)(int)trunc; the sum wraps
                ny = (py + (int(oy) & 0xFFFFFFFF)) & 0xFFFFFFFF
                nx, ny = min(nx, W - 1), min(ny, H - 1)
                nb = temporal[ny][nx]
                ph = lights.target_function(nb["Y"], hit)
                update_reservoir(rng, nr, nb["Y"], f32(f32(ph * nb["W_X"]) * f32(nb["M"])), nb["M"])
            if nr["Y"] != -1:
                finalize_resampling(nr, lights.target_function(nr["Y"], hit))
            out[py][px] = nr
    return out


RES = np.dtype([("Y", "<i4"), ("W_sum", "<f4"), ("W_X", "<f4"), ("M", "<i4")])


def to_array(rs):
    H, W = len(rs), len(rs[0])
    a = np.zeros((H, W), dtype=RES)
    for y in range(H):
        for x in range(W):
            r = rs[y][x]
            a[y, x] = (r["Y"], r["W_sum"], r["W_X"], r["M"])
    return a


def main():
    rng = np.random.default_rng(20261004)
    out = {}
    # ---- closest-hit cases: one single-triangle mesh each
    N = 320
    perms = [np.eye(3, dtype=f32), np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]], dtype=f32), np.array([[-1, 0, 0], [0, 1, 0], [0, 0, -1]], dtype=f32),
             np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0]], dtype=f32)]
    cases = dict(normals=np.zeros((N, 3, 3), f32), attribs=np.zeros((N, 2), f32), o2w=np.zeros((N, 3, 3), f32), w2o=np.zeros((N, 3, 3), f32), ray_dir=np.zeros((N, 3), f32),
                 mtype=np.zeros(N, i32), prop=np.zeros(N, f32), base=np.zeros((N, 3), f32), seed_in=np.zeros(N, u32), color=np.zeros((N, 3), f32), scatter=np.zeros((N, 3), f32),
                 scattered=np.zeros(N, i32), normal=np.zeros((N, 3), f32), seed_out=np.zeros(N, u32))
    for k in range(N):
        n = rng.normal(size=(3, 3)).astype(f32)
        n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(f32)
        if k % 5 == 0:
            n[1], n[2] = n[0], n[0]  # flat shading
        u = f32(rng.random() * 0.9)
        v = f32(rng.random() * (0.95 - float(u)))
        scale = f32(2.0 ** int(rng.integers(-2, 3)))
        R = perms[k % 4]
        o2w = (R * scale).astype(f32)                 # rotation-by-permutation times a power-of-two scale
        w2o = (R.T / scale).astype(f32)               # its inverse: exact
        d = rng.normal(size=3).astype(f32)
        if k % 7 == 0:
            d = (d * f32(3.7)).astype(f32)            # un-normalised directions (rgen:61 hands scatterDirection on as it is)
        mtype = int(k % 4) if k % 4 != 3 else 3
        prop = f32([0.0, rng.random() * 0.5, 1.0 + rng.random(), 0.0][k % 4])
        if k % 4 == 2 and k % 8 == 2:
            prop = f32(1.5)
        base = rng.random(3).astype(f32)
        seed = u32(rng.integers(0, 2 ** 32))
        color, scatter, scattered, wn, seed_out = closest_hit(n[0], n[1], n[2], np.array([u, v], dtype=f32), w2o, d, mtype, prop, base, seed)
        for name, val in (("normals", n), ("attribs", (u, v)), ("o2w", o2w), ("w2o", w2o), ("ray_dir", d), ("mtype", mtype), ("prop", prop), ("base", base), ("seed_in", seed),
                          ("color", color), ("scatter", scatter), ("scattered", int(scattered)), ("normal", wn), ("seed_out", seed_out)):
            cases[name][k] = val
    for k, v in cases.items():
        out["rchit_" + k] = v
    # ---- ReSTIR chains: small frames, synthetic G-buffers, three consecutive frames each
    chains = []
    for ci, (W, H, nl, max_used) in enumerate([(12, 9, 6, 6), (16, 10, 40, 24), (9, 7, 3, 1000)]):
        pos = (rng.random((nl, 3)) * 8 - 4).astype(f32)
        inten = (rng.random((nl, 3)) * 3 + 0.1).astype(f32)
        lights = Lights(pos, inten, nl, max_used)
        g = np.zeros((H, W, 4), dtype=f32)
        g[..., :3] = (rng.random((H, W, 3)) * 6 - 3).astype(f32)
        g[..., 3] = 1
        g[0, 0] = (1, 1, 1, 0)  # a cleared texel (the G-buffer's clear colour, pass.rs:210-214)
        # a projection-view matrix that sends most positions into the frame (column-major)
        pv = np.zeros(16, dtype=f32)
        pv[0], pv[5], pv[10], pv[14], pv[11], pv[15] = 0.2, 0.25, -1.0, -0.1, -0.05, 1.0
        hist = [[new_reservoir() for _ in range(W)] for _ in range(H)]
        for y in range(H):
            for x in range(W):
                hist[y][x] = {"Y": int(rng.integers(-1, nl)), "W_sum": f32(rng.random() * 2), "W_X": f32(rng.random()), "M": int(rng.integers(0, 30))}
        frames = []
        prev = hist
        for fi in range(3):
            frame = 1 + fi
            ini = initial_ris(lights, g, W, H, frame)
            tem = temporal_reuse(lights, g, W, H, frame, ini, prev, pv, enabled=(ci != 2 or fi != 1))
            spa = spatial_reuse(lights, g, W, H, frame, tem, enabled=(ci != 1 or fi != 2))
            frames.append((to_array(ini), to_array(tem), to_array(spa)))
            prev = [[dict(r) for r in row] for row in spa]
        chains.append(dict(W=W, H=H, pos=pos, inten=inten, max_used=max_used, g=g, pv=pv, hist=to_array(hist), frames=frames,
                           temporal_on=[int(ci != 2 or fi != 1) for fi in range(3)], spatial_on=[int(ci != 1 or fi != 2) for fi in range(3)]))
    out["chains"] = np.int32(len(chains))
    for ci, c in enumerate(chains):
        p = f"chain{ci}_"
        out[p + "size"] = np.int32([c["W"], c["H"], c["max_used"]])
        for name in ("pos", "inten", "g", "pv", "hist"):
            out[p + name] = c[name]
        out[p + "temporal_on"], out[p + "spatial_on"] = np.int32(c["temporal_on"]), np.int32(c["spatial_on"])
        for fi, (a, b, d) in enumerate(c["frames"]):
            out[p + f"f{fi}_initial"], out[p + f"f{fi}_temporal"], out[p + f"f{fi}_spatial"] = a, b, d
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", N, "closest-hit cases,", len(chains), "reservoir chains of 3 frames")


if __name__ == "__main__":
    main()
