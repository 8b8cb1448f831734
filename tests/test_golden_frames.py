"""Committed golden frames (tests/golden/cornell_frames.npz, generator alongside): the oracle must
reproduce them bit for bit on any host (CPU), and so must the HIP path through the C ABI (GPU)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_oracle_frames as gold  # noqa: E402

import oracle_api as oa  # noqa: E402
import rust_renderer_amd as rr  # noqa: E402

GOLDEN = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_frames.npz"))


def check(data):
    for k in GOLDEN.files:
        a, b = GOLDEN[k], np.asarray(data[k])
        if a.dtype.kind == "f":
            assert np.array_equal(a.view(np.uint32), b.astype(a.dtype).view(np.uint32)), f"{k} differs from the golden fixture"
        else:
            assert np.array_equal(a, b), f"{k} differs from the golden fixture"


def test_oracle_reproduces_golden_frames():
    check(gold.render(oa.OracleRenderer(gold.W, gold.H, threads=3)))


def test_oracle_brute_force_reproduces_golden_frames():
    check(gold.render(oa.OracleRenderer(gold.W, gold.H, brute_force=True)))


@pytest.mark.gpu
def test_hip_path_reproduces_golden_frames():
    check(gold.render(rr.Renderer(gold.W, gold.H)))
