"""bench.py's roofline block on fake statistics (no GPU): the per-FRAME HBM fraction must not depend on how the timed
steps were cut into wavefronts (--steps 20 = a 16-frame and a 4-frame wavefront read 0.33, --steps 64
read 0.59 for the same frames), the bound comes from the counter profile, and the metric names the real step count."""
import importlib.util
import json
import os
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("uh_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


PROFILE = {
    "signature": {"config": 1},
    "source": "test profile",
    "frame_hbm_bytes": 9.27e9,
    "frame_hbm_bytes_uncorrected": 5.16e9,
    "rays_per_frame": 17.79e6,
    "kernels": {
        "k_trace_closest": {"launch_ns": 3.667e6, "hbm_bytes_per_launch": 14.09e9, "hbm_bytes_per_launch_uncorrected": 7.51e9, "closest_rays_per_launch": 29.45e6,
                            "issue_frac": 0.57, "lane_utilisation": 0.61, "ta_busy_frac": 0.92, "td_busy_frac": 0.96, "wave_instr_per_launch": 2.36e9},
        "k_shade_hit": {"launch_ns": 1.69e6, "hbm_bytes_per_launch": 10.87e9},
    },
}


def fake_run(bench, steps, prof=PROFILE, ms_per_step=2.15):
    """the statistics a run of `steps` frames leaves: 5 bounces per wavefront, wavefronts of up to 16 frames"""
    closest_per_frame, rays_per_frame = 5 * 29.45e6 / 16, 17.79e6  # a launch = one bounce of one wavefront
    wavefronts = (steps + 15) // 16
    st = types.SimpleNamespace(trace_closest_launches=5 * wavefronts, trace_closest_ms=5 * wavefronts * 4.0, trace_shadow_ms=1.0, shade_ms=1.0)
    args = types.SimpleNamespace(steps=steps)
    return bench.roofline(args, st, alone_ms=3.62, alone_rays=29.45e6, my_closest=closest_per_frame * steps, nodes_per_ray=18.8, tris_per_ray=3.3,
                          elapsed=steps * ms_per_step * 1e-3, sig={"config": 1}, rays_per_frame=rays_per_frame, prof=prof)


def test_frame_fraction_does_not_depend_on_the_step_count(bench):
    r20, r64 = fake_run(bench, 20), fake_run(bench, 64)
    want = 9.27e9 / 2.15e-3 / 8e12
    assert r20["frame_hbm_frac"] == pytest.approx(want, rel=1e-9)
    assert r64["frame_hbm_frac"] == pytest.approx(want, rel=1e-9)
    assert r20["uncorrected"]["frame_hbm_frac"] == pytest.approx(r64["uncorrected"]["frame_hbm_frac"], rel=1e-9)
    # a LAUNCH's traffic does go with the rays it carries: 20 steps = 2 wavefronts of 10 frames on average
    assert r20["all_reads_doubled"]["traffic"] == pytest.approx(14.09e9 * (10 / 16), rel=1e-9)
    assert r64["all_reads_doubled"]["traffic"] == pytest.approx(14.09e9, rel=1e-9)
    # the calibrated figure: the counters as reported + half of what the walk streams (32 bytes per ray); scattered record reads are
    # reported exactly (tools/microbench/fetch_size.hip)
    assert r64["traffic"] == pytest.approx(7.51e9 + 0.5 * 32.0 * 29.45e6, rel=1e-9)
    assert r20["traffic"] == pytest.approx((7.51e9 + 0.5 * 32.0 * 29.45e6) * (10 / 16), rel=1e-9)
    assert r64["uncorrected"]["frac"] < r64["frac"] < r64["all_reads_doubled"]["frac"]
    # ... and the serialised fraction is the same for both (bytes and time scale together)
    assert r20["frac"] == pytest.approx(r64["frac"], rel=1e-9)
    assert 0.0 < r64["frac"] < 1.0


def test_frame_bytes_follow_the_rays_of_a_frame_only(bench):
    prof = json.loads(json.dumps(PROFILE))
    prof["rays_per_frame"] = 2 * 17.79e6  # the profiled frames carried twice the rays: a frame of this run moves half the bytes
    r = fake_run(bench, 20, prof)
    assert r["frame_hbm_bytes"] == pytest.approx(9.27e9 / 2, rel=1e-9)
    del prof["rays_per_frame"]  # an old profile without the figure: no scaling at all
    assert fake_run(bench, 20, prof)["frame_hbm_bytes"] == pytest.approx(9.27e9, rel=1e-9)


def test_bound_is_read_from_the_counters(bench):
    r = fake_run(bench, 64)
    assert r["bound"].startswith("vmem-issue")  # texture addresser 0.92 busy, HBM 0.48, VALU 0.57
    assert "0.92" in r["bound_note"] and "0.57" in r["bound_note"]
    prof = json.loads(json.dumps(PROFILE))
    k = prof["kernels"]["k_trace_closest"]
    k["ta_busy_frac"], k["issue_frac"], k["hbm_bytes_per_launch"], k["hbm_bytes_per_launch_uncorrected"] = 0.3, 0.2, 47e9, 24e9  # 24.5 GB (calibrated) in 3.667 ms = 0.83 of the peak
    assert fake_run(bench, 64, prof)["bound"] == "hbm"
    k["issue_frac"] = 0.8
    assert fake_run(bench, 64, prof)["bound"] == "hbm/valu"
    none = fake_run(bench, 64, prof=None)
    assert none["bound"] is None and none["traffic"] is None and none["frame_hbm_frac"] is None


def test_committed_profile_reads_the_same_fraction_at_20_and_64_steps(bench):
    doc = json.load(open(os.path.join(ROOT, "profiles", "bench_counters.json")))
    profiles = doc.get("profiles", [doc])
    sigs = [json.dumps(p["signature"], sort_keys=True) for p in profiles]
    assert len(set(sigs)) == len(sigs), "one profile per bench command line"
    prof = next(p for p in profiles if p["signature"]["config"] == 1 and not p["signature"]["opts"])  # the bench line's own
    a, b = fake_run(bench, 20, prof), fake_run(bench, 64, prof)
    assert a["frame_hbm_frac"] == pytest.approx(b["frame_hbm_frac"], rel=1e-9)
    assert a["frame_hbm_frac"] == pytest.approx(prof["frame_hbm_bytes"] * (17.79e6 / prof["rays_per_frame"] if prof.get("rays_per_frame") else 1.0) / 2.15e-3 / 8e12, rel=1e-9)
    for p in profiles:  # every profile names the kernels the roofline block may pick
        assert "k_trace_closest" in p["kernels"], p["signature"]


def test_metric_names_the_real_step_count():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "64 frames x 1 spp" not in src
    assert "{args.steps} frames x {args.spp} spp" in src
    for key in ('"value_tree_walk"', '"sun_grid"', '"camera_grid"', '"value_with_grid_builds"', '"rccl_library_comm_ranks"', '"hip"'):
        assert key in src, key
    assert "camera and sun at rest" in src and "outside the timed region" in src  # the headline says what regime it is measured in
    assert "import torch" not in src, "a rank is a GPU process and imports no torch"
