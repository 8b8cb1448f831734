"""-m gpu: every host verb that touches device state, interleaved with four frames in flight, against the same call sequence
with a uh_synchronize before and after the verb (include/utopian_hip.h "Stream ordering": each of these verbs waits for the
frames in flight and is complete when it returns, whatever streams the library spreads its frames over).

Two of them were races until round 3's last commits and were found by a soak tool, not by the suite:
  46b4f3d  uh_reset_stats cleared the counters with a hipMemset on the null stream, which the frames' non-blocking streams do
           not wait for: the clear could land after the next frame's first counter updates    -> case "reset_stats"
  19abe93  uh_reset_accumulation enqueued its clears on slot 0's stream and returned; the next frame may run on another
           slot's stream, which does not wait for them                                          -> case "reset_accumulation"
Each case runs the in-flight form several times: a race shows as a mismatch in some repetition, not in every one."""
import numpy as np
import pytest

import rust_renderer_amd as rr
from rust_renderer_amd.api import transform3x4
from rust_renderer_amd.types import RESERVOIR_DTYPE

pytestmark = pytest.mark.gpu

W, H = 96, 64
REPEATS = 6


def seeded_reservoirs():
    r = np.zeros((H, W), dtype=RESERVOIR_DTYPE)
    k = np.arange(W * H).reshape(H, W)
    r["Y"], r["W_sum"], r["W_X"], r["M"] = k % 3, 0.25 + (k % 7) * 0.125, 0.5 + (k % 5) * 0.25, 1 + k % 4
    return r


def v_reset_stats(r, loop):
    r.reset_stats()


def v_reset_accumulation(r, loop):
    loop.reset()  # uh_reset_accumulation + total_samples = 0 (main.rs:400-413)


def v_write_reservoirs(r, loop):
    r.write_reservoirs(2, seeded_reservoirs())


def v_options(r, loop):
    for k, v in (("furnace", 1), ("overlap", 0), ("sun_grid", 0), ("frames_in_flight", 2), ("batch_frames", 1), ("count_visits", 1), ("time_kernels", 1),
                 ("time_kernels", 0), ("trace_blocks_per_cu", 3), ("full_frame_restir", 1)):
        r.set_option(k, v)


def v_tile_partition(r, loop):
    r.set_tile_partition(1, 2, 16)


def v_restir_partition(r, loop):
    r.set_restir_partition(0, 2)  # this context's band of rows from now on (no exchange: the other band stays as it is)


def v_queries(r, loop):
    r.get_stats()
    r.read_accumulation()
    r.read_output_bgra8()
    r.read_reservoirs(2)
    rays = np.float32([[0.0, 0.9, 2.0, 0.001, 0.0, -0.2, -1.0, 10000.0], [0.0, 0.9, 2.0, 0.001, 0.3, 0.1, -1.0, 10000.0]])
    r.trace_closest(rays)
    r.trace_any(rays)


def v_upload_and_rebuild(r, loop):
    r.add_texture(np.full((8, 8, 4), 200, dtype=np.uint8))
    r.initialize_raytracing()  # uh_build_acceleration: new node / packet / table uploads while frames still traverse the old ones


def v_move_and_refit(r, loop):
    r.set_instance_transform(6, transform3x4((0.3,) * 3, (-0.35, 0.4, -0.2)))
    r.refit_acceleration()


def v_device_rebuild(r, loop):
    r.set_option("device_build", 1)
    r.initialize_raytracing()


CASES = {"reset_stats": v_reset_stats, "reset_accumulation": v_reset_accumulation, "write_reservoirs": v_write_reservoirs, "options": v_options,
         "tile_partition": v_tile_partition, "restir_partition": v_restir_partition, "queries": v_queries, "upload_and_rebuild": v_upload_and_rebuild,
         "move_and_refit": v_move_and_refit, "device_rebuild": v_device_rebuild}


def run(scene, verb, serial, batched_tail):
    r = scene.upload(rr.Renderer(W, H, device=0))
    loop = rr.FrameLoop(r, scene.make_view(W, H))
    for _ in range(4):
        loop.frame(rr.PASS_ALL)  # four frames in flight: one per slot, their reservoir chains on the reservoir stream
    if serial:
        r.synchronize()
    verb(r, loop)
    if serial:
        r.synchronize()
    if batched_tail:
        loop.frames(5, rr.PASS_ALL)
    else:
        for _ in range(5):
            loop.frame(rr.PASS_ALL)
    s = r.get_stats()
    out = (r.read_accumulation().copy(), [r.read_reservoirs(k).copy() for k in range(3)], list(s.rays), int(s.closest_hits), int(s.misses), int(s.frames))
    return out


@pytest.fixture(scope="module")
def cornell():
    return rr.scenes.cornell_scene(subdivisions=2, tex_size=16)


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("batched_tail", [False, True])
def test_verb_between_frames_in_flight_equals_the_serial_sequence(cornell, name, batched_tail):
    want = run(cornell, CASES[name], True, batched_tail)
    for rep in range(REPEATS):
        got = run(cornell, CASES[name], False, batched_tail)
        assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), f"{name}: accumulation differs (repetition {rep})"
        for k in range(3):
            assert np.array_equal(got[1][k], want[1][k]), f"{name}: reservoir buffer {k} differs (repetition {rep})"
        assert got[2:] == want[2:], f"{name}: counters differ (repetition {rep}): {got[2:]} != {want[2:]}"
