#!/usr/bin/env python3
"""Generates tests/golden/cornell_frames.npz from the CPU oracle (oracle/oracle.cpp): the
Cornell-class scene (all four material types, textures, 3 lights), 40x32, 3 frames of the full pass
chain (G-buffer cast, ReSTIR reset/initial/temporal/spatial, path tracing) with the sky disabled so
that every value is produced by correctly rounded IEEE f32 arithmetic only (no libm exp/pow in the
radiance) and is therefore reproducible bit for bit on any host and on the GPU.

The reference renderer cannot run in this environment; these vectors pin
THIS repo's oracle against regressions and give the HIP path a committed target.
  python tests/golden/make_oracle_frames.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import oracle_api as oa  # noqa: E402
import rust_renderer_amd as rr  # noqa: E402

W, H, FRAMES = 40, 32, 3


def scene():
    return rr.scenes.cornell_scene(subdivisions=1, tex_size=8)


def render(renderer):
    sc = scene()
    sc.upload(renderer)
    loop = rr.FrameLoop(renderer, sc.make_view(W, H, sky_enabled=0, use_ris_light_sampling=1))
    for _ in range(FRAMES):
        loop.frame(rr.PASS_ALL)
    s = renderer.get_stats()
    out = dict(
        accumulation=renderer.read_accumulation(),
        gbuffer_position=renderer.read_gbuffer_position(),
        rays=np.array(list(s.rays), dtype=np.uint64),
        closest_hits=np.uint64(s.closest_hits),
        misses=np.uint64(s.misses),
    )
    for i, name in enumerate(("initial", "temporal", "spatial")):
        r = renderer.read_reservoirs(i)
        for f in ("Y", "W_sum", "W_X", "M"):
            out[f"reservoir_{name}_{f}"] = r[f].copy()
    return out


if __name__ == "__main__":
    data = render(oa.OracleRenderer(W, H, threads=2))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cornell_frames.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes; rays", data["rays"].tolist())
