#!/usr/bin/env python3
"""Makes tests/golden/sky_kat.json: known-answer vectors for the sky of the miss shader (reference.rmiss:16-23:
min(IntegrateScattering(origin, dir, 999999999, normalize(sun), 1), 1)), computed by a numpy restatement of
utopian/shaders/include/atmosphere.glsl:53-214 WRITTEN FROM THE GLSL TEXT - independently of oracle/oracle.cpp and of
csrc/device_math.h, which the fixture pins. Two evaluations per vector:
  f64  the mathematical value (every operation in float64);
  f32  every operation in float32, in the shader's order - what a GPU running the GLSL computes, up to its exp / pow. The two
       differ by up to ~1e-3 relative: AtmosphereHeight subtracts the planet radius (6,371,000) from a distance of the same
       size, so a float32 height carries +-0.5 m, i.e. +-4e-4 of the Mie scale height (1,200 m) - a property of the shader.
The oracle (float32, libm) must match f32 closely and f64 within that cancellation error."""
import json
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sky_kat.json")


def make(T):
    """the atmosphere functions with every constant and intermediate in dtype T (np.float64 or np.float32)"""
    PI = T(3.14159265359)
    PLANET_RADIUS = T(6371000)
    PLANET_CENTER = np.array([0, -6371000, 0], dtype=T)
    ATMOSPHERE_HEIGHT = T(100000)
    RAYLEIGH_HEIGHT = ATMOSPHERE_HEIGHT * T(0.08)
    MIE_HEIGHT = ATMOSPHERE_HEIGHT * T(0.012)
    C_RAYLEIGH = np.array([5.802, 13.558, 33.100], dtype=T) * T(1e-6)
    C_MIE = np.array([3.996, 3.996, 3.996], dtype=T) * T(1e-6)
    C_OZONE = np.array([0.650, 1.881, 0.085], dtype=T) * T(1e-6)
    EXPOSURE = T(20)

    def dot(a, b):
        return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]

    def sphere_intersection(start, d, center, radius):  # :53-69
        s = start - center
        a = dot(d, d)
        b = T(2.0) * dot(s, d)
        c = dot(s, s) - (radius * radius)
        disc = b * b - T(4) * a * c
        if disc < 0:
            return T(-1), T(-1)
        disc = np.sqrt(disc)
        return (-b - disc) / (T(2) * a), (-b + disc) / (T(2) * a)

    def atmosphere_intersection(start, d):  # :74-77
        return sphere_intersection(start, d, PLANET_CENTER, PLANET_RADIUS + ATMOSPHERE_HEIGHT)

    def phase_rayleigh(costh):  # :81-84
        return T(3) * (T(1) + costh * costh) / (T(16) * PI)

    def phase_mie(costh, g):  # :85-91
        g = min(g, T(0.9381))
        k = T(1.55) * g - T(0.55) * g * g * g
        kcosth = k * costh
        return (T(1) - k * k) / ((T(4) * PI) * (T(1) - kcosth) * (T(1) - kcosth))

    def height(p):  # :95-98
        q = p - PLANET_CENTER
        return np.sqrt(dot(q, q)) - PLANET_RADIUS

    def density(h):  # :99-115
        r = np.exp(-max(T(0), h / RAYLEIGH_HEIGHT))
        m = np.exp(-max(T(0), h / MIE_HEIGHT))
        o = max(T(0), T(1) - abs(h - T(25000.0)) / T(15000.0))
        return np.array([r, m, o], dtype=T)

    def optical_depth(start, d):  # :123-143
        _, ray_length = atmosphere_intersection(start, d)
        n = 8
        step = ray_length / T(n)
        od = np.zeros(3, dtype=T)
        for i in range(n):
            p = start + d * (T(i) + T(0.5)) * step
            od = od + density(height(p)) * step
        return od

    def absorb(od):  # :146-150
        return np.exp(-(od[0] * C_RAYLEIGH + od[1] * C_MIE * T(1.1) + od[2] * C_OZONE) * T(1))

    def integrate_scattering(start, d, ray_length, light_dir):  # :154-214, lightColor = 1
        ray_height = height(start)
        exponent = T(1) + min(max(T(1) - ray_height / ATMOSPHERE_HEIGHT, T(0)), T(1)) * T(8)
        i0, i1 = atmosphere_intersection(start, d)
        ray_length = min(ray_length, i1)
        if i0 > 0:
            start = start + d * i0
            ray_length = ray_length - i0
        costh = dot(d, light_dir)
        phase_r, phase_m = phase_rayleigh(costh), phase_mie(costh, T(0.85))
        n = 16
        od = np.zeros(3, dtype=T)
        rayleigh = np.zeros(3, dtype=T)
        mie = np.zeros(3, dtype=T)
        prev = T(0)
        for i in range(n):
            ray_time = np.power(T(i) / T(n), exponent) * ray_length
            step = ray_time - prev
            p = start + d * ray_time
            dens = density(height(p))
            od = od + dens * step
            view_t = absorb(od)
            light_t = absorb(optical_depth(p, light_dir))
            rayleigh = rayleigh + view_t * light_t * phase_r * dens[0] * step
            mie = mie + view_t * light_t * phase_m * dens[1] * step
            prev = ray_time
        return (rayleigh * C_RAYLEIGH + mie * C_MIE) * T(1) * EXPOSURE

    def miss(origin, direction, sun):  # reference.rmiss:16-23
        o, d, s = (np.array(v, dtype=T) for v in (origin, direction, sun))
        s = s * (T(1) / np.sqrt(dot(s, s)))  # normalize(view.sun_dir)
        with np.errstate(over="ignore", invalid="ignore"):
            c = integrate_scattering(o, d, T(999999999.0), s)
        return c, np.minimum(c, T(1))

    return miss


def main():
    rng = np.random.default_rng(20261004)
    origins = [(0, 0, 0), (-10.28, 2.1, -0.18), (12.5, 11.9, -6.0), (0, 1000.0, 0), (300.0, 25000.0, -40.0), (0.0, 99000.0, 0.0), (0.0, 150000.0, 0.0), (5e5, 2.0e5, -3e5)]
    suns = [(0.0, 0.9, 0.15), (0.3, 0.1, -0.9), (0.0, 1.0, 0.0), (-0.6, 0.02, 0.4), (0.2, -0.3, 0.5)]
    dirs = [(0, 1, 0), (1, 0, 0), (0, 0, -1), (0.7, 0.05, 0.7), (0.0, -0.2, 1.0), (0.0, 0.9, 0.15), (-0.3, 0.4, 0.2), (0.0, -1.0, 0.0)]
    vectors = []
    for i in range(48):
        o = origins[i % len(origins)]
        s = suns[(i // 3) % len(suns)]
        d = dirs[(i * 5 + i // 8) % len(dirs)] if i < 32 else tuple(rng.normal(size=3))
        if i >= 40:  # un-normalised directions, as bounce rays carry them (reference.rgen:61)
            d = tuple(np.array(d) * rng.uniform(0.3, 2.5))
        elif i >= 8:
            d = tuple(np.array(d, dtype=np.float64) / np.linalg.norm(d))
        vectors.append((tuple(float(np.float32(x)) for x in o), tuple(float(np.float32(x)) for x in d), tuple(float(np.float32(x)) for x in s)))
    for s in suns[:4]:  # straight into the sun (Mie forward peak: the clamp of reference.rmiss:22), from the ground and from 1 km
        n = np.array(s, dtype=np.float64) / np.linalg.norm(s)
        for o in origins[1], origins[3]:
            vectors.append((tuple(float(np.float32(x)) for x in o), tuple(float(np.float32(x)) for x in n), tuple(float(np.float32(x)) for x in s)))
    f64, f32 = make(np.float64), make(np.float32)
    out = []
    for o, d, s in vectors:
        raw64, sky64 = f64(o, d, s)
        raw32, sky32 = f32(o, d, s)
        out.append(dict(origin=o, direction=d, sun=s, unclamped_f64=[float(x) for x in raw64], unclamped_f32=[float(x) for x in raw32], sky_f64=[float(x) for x in sky64],
                        sky_f32=[float(x) for x in sky32]))
    json.dump(dict(source="numpy restatement of utopian/shaders/include/atmosphere.glsl:53-214 + reference.rmiss:16-23 (tests/golden/make_sky_fixture.py)", vectors=out),
              open(OUT, "w"), indent=1)
    rel = max(np.abs(np.array(v["sky_f64"]) - np.array(v["sky_f32"])).max() / max(np.abs(np.array(v["sky_f64"])).max(), 1e-9) for v in out)
    print(len(out), "vectors ->", OUT, "; largest f32-vs-f64 relative difference", rel)


if __name__ == "__main__":
    main()
