"""-m gpu: the multi-RANK path of `bench.py --gpus N` end to end on a one-GPU box - N rank processes on GPU 0, rendezvous, uh_rccl_attach,
the reservoir bands' all-gather and uh_rccl_gather_tiles inside the library - with a test double for librccl (tests/cpp/fake_rccl.cpp:
RCCL itself refuses two ranks on one GPU). What it proves is the library's own side of a multi-rank job - who packs what, who sends
and receives how much in which order, what the root composes, what the bands exchange - on real device buffers: the composed frame
must equal one context's, bit for bit. RCCL's side (the transport over xGMI) is the driver's 8-GPU run to show."""
import os
import subprocess
import sys

import numpy as np
import pytest

import rust_renderer_amd as rr
from util import run_frames

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake_rccl_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("fake_rccl")
    subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp"),
                    "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-o", str(d / "librccl.so.1")], check=True)
    return str(d)


@pytest.mark.parametrize("world,lights,tile", [(2, 0, 16), (3, 1, 8), (4, 1, 32)])
def test_rank_processes_compose_one_contexts_frame(fake_rccl_dir, tmp_path, world, lights, tile):
    W, H, frames = 160, 90, 3
    out = str(tmp_path / "composed.npz")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = fake_rccl_dir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    rc = rr.launch.spawn_ranks(world, os.path.join(ROOT, "tests", "rehearsal_worker.py"), [out, str(W), str(H), str(tile), str(frames), str(lights)], env=env, timeout=300)
    assert rc == 0
    got = np.load(out)
    scene = rr.scenes.sponza_class_scene(detail=0.12, tex_size=32, with_spheres=True, num_lights=64 if lights else 0, sphere_subdivisions=2)
    ref = scene.upload(rr.Renderer(W, H))
    run_frames(ref, scene, W, H, frames, rr.PASS_ALL if lights else rr.PASS_REFERENCE_PT, use_ris_light_sampling=1 if lights else 0)
    assert np.array_equal(got["acc"].view(np.uint32), ref.read_accumulation().view(np.uint32))
    assert np.array_equal(got["bgra"], ref.read_output_bgra8())
    if lights:
        assert np.array_equal(got["spatial"], ref.read_reservoirs(2))
    # the ranks' path rays add up to the one context's (the reservoir passes' G-buffer rays are cast per band)
    total = np.load(out + ".rays.npy")
    want = np.array(list(ref.get_stats().rays), dtype=np.int64)
    assert list(total[:4]) == list(want[:4])


@pytest.mark.parametrize("config,gpus", [(1, 2), (2, 2), (1, 4)])
def test_bench_gpus_n_on_one_gpu(fake_rccl_dir, config, gpus):
    """`bench.py --gpus N` as the driver launches it (bench.py becomes the launcher of its own ranks), every rank on GPU 0 (config 2: the reservoir
    passes by bands of rows, one all-gather per frame): the rendezvous,
    the barrier-bracketed timed region, the per-step composition through uh_rccl_gather_tiles, the max-over-ranks time and rank 0's one JSON line"""
    import json

    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = fake_rccl_dir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--config", str(config), "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-tree-walk", "--no-alone",
                        "--width", "640", "--height", "360", "--rank-device", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE json line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["rccl_library_comm_ranks"] == gpus and d["steps"] == 8 and d["warmup"] == 2
    assert d["scaling"] == "weak" or d["scaling"] == "strong"
    assert d["value"] > 0 and d["ms_per_step"] > 0
