#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: Mrays/s and ms/frame on the Sponza-class 1080p path trace.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
  or plainly (python bench.py --gpus N): the script then starts its N ranks itself, as child processes under
  torch.distributed.run on 127.0.0.1, before anything has touched a GPU, and exits with their status.
  A rank is a GPU process and imports no torch: the ranks find each other through rust-renderer_amd/launch.py's Rendezvous
  (TCP on 127.0.0.1), which hands the ncclUniqueId round, and every collective of the data path is RCCL inside the library.

A step = one frame = one pass of reference_pt_pass over the 1920x1080 framebuffer at 1 sample per
pixel and 5 bounces (the reference's per-frame dispatch, renderers/mod.rs:357; 64 steps = the
64 spp headline image). The scene, BVH, textures and all path state are resident in HBM before the
timed region. N > 1: the framebuffer is tile-partitioned (64x64 tiles, round-robin) — each rank
traces only its tiles, and ONE RCCL gather of the RGBA32F accumulation tiles to rank 0 (inside the
timed region, after the last step; uh_rccl_gather_tiles, enqueued behind the frames) composes the final image. Total work is
fixed => "strong".

Prints ONE JSON line (rank 0). `roofline` is for the dominant kernel, k_trace_closest, timed live
with HIP events on the library's own stream; `cpu_baseline` is the CPU oracle on this box's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _load_launch():
    """rust-renderer_amd/launch.py by path: no torch, no HIP, not even the package - safe before any GPU is touched"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("uh_launch", os.path.join(ROOT, "rust-renderer_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


launch = _load_launch()

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", type=int, default=1, help="BASELINE.json configs index: 1 = Sponza-class diffuse (headline), 2 = + 1024 lights ReSTIR")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tree-walk", action="store_true", help="skip the second timed run with both grids off (profiling runs: every frame in the counters is then a frame of the timed kind)")
    ap.add_argument("--no-alone", action="store_true", help="skip the untimed standalone-kernel calibration frames (profiling runs: every launch in the trace is then in the timed regime)")
    ap.add_argument("--emulate-world", type=int, default=0, help="single process: trace only one rank's tiles of an N-rank partition (what one rank sees at --gpus N)")
    ap.add_argument("--emulate-rank", type=int, default=0, help="the rank --emulate-world stands in for")
    ap.add_argument("--force-dist", action="store_true", help="run the rendezvous + RCCL composition path even with one rank (rehearsal on a 1-GPU box)")
    ap.add_argument("--rank-device", type=int, default=None, help="every rank on this device instead of its LOCAL_RANK's (rehearsal of N rank processes on a 1-GPU box: tests/test_gpu_rehearsal.py)")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (experiments), e.g. sun_grid=0")
    ap.add_argument("--full-frame-reservoir-passes", action="store_true", help="N > 1: every rank runs the G-buffer cast and the reservoir passes for the whole frame (rounds 1-2) instead of its band of rows + one all-gather per frame")
    ap.add_argument("--spp", type=int, default=1, help="samples_per_frame (reference default 1, UI maximum 10); SURVEY 8d also asks for 64 spp as 8 frames x 8")
    ap.add_argument("--cook-torrance", action="store_true", help="extension (SURVEY 8f N2): the diffuse materials of configs 1-3 become Cook-Torrance (material type 4)")
    ap.add_argument("--cpu-sample", type=str, default="1920x1080x8", help="WxHx(max frames) rendered by the CPU oracle")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="the CPU baseline stops after the first frame that ends beyond this many seconds")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and not launch.under_launcher():
        # plain `python bench.py --gpus N`: become the launcher. Nothing has initialised a GPU yet (no torch, no
        # library call); the N ranks are fresh child processes, this one only waits for them.
        sys.exit(launch.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if world != args.gpus:
        args.gpus = world

    import numpy as np

    import rust_renderer_amd as rr  # binds libutopian_hip.so to /opt/rocm's HIP runtime, the one it was built with (api._preload_hip_runtime)

    use_dist = world > 1 or args.force_dist
    rdzv = None
    if use_dist:
        rdzv = launch.Rendezvous.from_env()
        print(f"[bench] rank {rank}/{world} joined the rendezvous", file=sys.stderr, flush=True)

    W, H = args.width, args.height
    kw = dict(tex_size=args.tex_size)
    if args.detail != 1.0 or args.config != 3:
        kw["detail"] = args.detail
    if args.cook_torrance and args.config in (1, 2, 3):
        kw["cook_torrance"] = True
    scene = rr.scenes.scene_for_config(args.config, **kw)
    renderer = rr.Renderer(W, H, device=local_rank if args.rank_device is None else args.rank_device)
    for kv in args.opt:
        k, v = kv.split("=")
        renderer.set_option(k, int(v))
    scene.upload(renderer)
    pass_mask = rr.PASS_ALL if args.config == 2 else rr.PASS_REFERENCE_PT
    restir = bool(pass_mask & rr.PASS_RESTIR) and not args.full_frame_reservoir_passes
    if use_dist:
        renderer.set_tile_partition(rank, world, args.tile)
        # one communicator per rank, inside the library (the 128-byte id goes round through the rendezvous): the tile gather of every
        # config, and for config 2 the reservoir passes by bands of rows with their all-gather per frame
        rr.distributed.attach_ranks(renderer, rdzv, band_partition=restir)
    elif args.emulate_world > 1:
        renderer.set_tile_partition(args.emulate_rank % args.emulate_world, args.emulate_world, args.tile)
        if restir:
            renderer.set_restir_partition(args.emulate_rank % args.emulate_world, args.emulate_world)  # one rank's rows, no exchange: its share of the work
    view = scene.make_view(W, H, samples_per_frame=args.spp) if args.spp != 1 else scene.make_view(W, H)
    loop = rr.FrameLoop(renderer, view)

    def sync_all():
        renderer.synchronize()  # every stream of the context idle - the composition's sends, receives and kernels included
        if use_dist:
            rdzv.barrier()

    def compose():
        # the ONE collective of the path tracer's data path: RCCL gather of the packed accumulation tiles to rank 0, which scatters
        # them and recomputes pt_output_image in one launch (uh_rccl_gather_tiles: enqueued behind the frames, no host wait)
        rr.distributed.gather_and_compose(renderer, rdzv, args.tile, resolve=(max(view.total_samples, 1), view.accumulation_limit))

    # ---- untimed: one counted frame gives nodes / triangles visited per ray of the dominant kernel (deterministic). A wavefront of
    # frames first: the sun grid and the camera grid exist from then on, as in the timed region - the primary rays then go through
    # k_trace_camera_grid, and k_trace_closest traces the bounce rays (and the primary rays of pixels with long lists)
    loop.frames(16, pass_mask)
    renderer.set_option("count_visits", 1)
    renderer.reset_stats()
    loop.frame(pass_mask)
    cs = renderer.get_stats()
    closest_rays = cs.rays[rr.RAY_BOUNCE] + (cs.camera_tree_rays if cs.camera_grid_cells else cs.rays[rr.RAY_PRIMARY])
    nodes_per_ray = cs.nodes_visited / max(closest_rays, 1)
    tris_per_ray = cs.tris_tested / max(closest_rays, 1)
    light_nodes_per_ray = cs.light_nodes_visited / max(cs.rays[rr.RAY_LIGHT_SHADOW], 1)
    light_tris_per_ray = cs.light_tris_tested / max(cs.rays[rr.RAY_LIGHT_SHADOW], 1)
    camera_tests_per_ray = cs.camera_grid_tris_tested / max(cs.rays[rr.RAY_PRIMARY] - cs.camera_tree_rays, 1) if cs.camera_grid_cells else None
    sun_rays_counted = max(cs.rays[rr.RAY_SUN_SHADOW], 1)
    sun_grid = {
        "in_use": bool(cs.sun_grid_cells),
        "build_ms": cs.sun_grid_build_ms,          # once per (geometry, sun direction), OUTSIDE the timed region: see value_with_sun_grid_build
        "cells": cs.sun_grid_cells,
        "entries": cs.sun_grid_entries,
        "bytes": cs.sun_grid_bytes,                # device memory of the structure (cells + lists + coarse cover + the 64-byte records when within sun_grid_inline_max_mb)
        "bytes_over_packet_array": cs.sun_grid_bytes / max(64.0 * scene.num_triangles, 1.0),
        "mean_list": cs.sun_grid_mean_list,
        "tests_per_ray": cs.shadow_tris_tested / sun_rays_counted,   # triangle tests per sun shadow ray (grid walk + the tree walk of the handed-over rays)
        "handed_to_tree": cs.sun_tree_rays / sun_rays_counted,       # share of the sun rays the grid gives back to the tree (border cells, long lists)
        "answered_by_cover": cs.sun_covered_rays / sun_rays_counted,  # share answered by the cell's cover depth alone
    }
    renderer.set_option("count_visits", 0)
    camera_grid = None  # filled after the timed region (the grid is built by the first multi-frame call)

    # ---- untimed: the dominant kernel alone on the GPU (no frame overlap, no second stream), in launches of the SAME size as
    # the timed ones (one wavefront of the library's default batch): its serialised launch duration, without co-scheduled kernels
    frames_rendered = 17  # the wavefront that builds the grids + the counted frame
    renderer.set_option("time_kernels", 1)
    serial = (("frames_in_flight", 1), ("overlap", 0))
    for k, v in serial:
        renderer.set_option(k, v)
    alone_ms = alone_rays = alone_light_ms = alone_light_rays = 0.0
    frame_by_frame_ms = serial_ms_per_frame = None
    if not args.no_alone:
        loop.frames(16, pass_mask)  # creates the slot; the same wavefront size as below
        renderer.reset_stats()
        loop.frames(16, pass_mask)
        alone = renderer.get_stats()
        alone_ms = alone.trace_closest_ms / max(alone.trace_closest_launches, 1)
        alone_rays = float(alone.rays[rr.RAY_BOUNCE] + (alone.camera_tree_rays if alone.camera_grid_cells else alone.rays[rr.RAY_PRIMARY])) / max(alone.trace_closest_launches, 1)
        alone_light_ms = alone.trace_light_ms / max(alone.trace_light_launches, 1)
        alone_light_rays = float(alone.rays[rr.RAY_LIGHT_SHADOW]) / max(alone.trace_light_launches, 1)
        serial_ms_per_frame = {"trace_closest": alone.trace_closest_ms / 16, "camera_grid": alone.camera_grid_ms / 16, "trace_shadow": alone.trace_shadow_ms / 16,
                               "trace_light": alone.trace_light_ms / 16, "shade_hit_and_miss": alone.shade_ms / 16}
        # what a caller of uh_render_frame sees with nothing overlapped: one frame per call, one stream, no batching
        # (per-kernel event timing is on in these frames: a few percent of launch overhead included)
        renderer.set_option("batch_frames", 1)
        renderer.synchronize()
        t_fbf = time.perf_counter()
        for _ in range(8):
            loop.frame(pass_mask)
        renderer.synchronize()
        frame_by_frame_ms = (time.perf_counter() - t_fbf) / 8 * 1e3
        frames_rendered += 40
    renderer.set_option("time_kernels", 0)
    for k, v in (("frames_in_flight", 4), ("overlap", 1), ("batch_frames", 0)):
        renderer.set_option(k, v)
    for kv in args.opt:
        k, v = kv.split("=")
        renderer.set_option(k, int(v))

    # ---- untimed: what an interactive caller sees with the library's defaults - uh_render_frame, then a synchronisation
    # (a present) after every frame: side-stream overlap inside the frame, but no second frame in flight and no batching
    interactive_frame_ms = pipelined_frame_ms = None
    if not args.no_alone:
        for _ in range(4):  # every frames-in-flight slot exists before the clock starts
            loop.frame(pass_mask)
        renderer.synchronize()
        t_int = time.perf_counter()
        for _ in range(8):
            loop.frame(pass_mask)
            renderer.synchronize()
        interactive_frame_ms = (time.perf_counter() - t_int) / 8 * 1e3
        # ... and with a swapchain's worth of frames in flight: uh_render_frame per frame (every frame may carry another
        # camera), no wait in between - the library rotates its four slots; no batching
        t_pipe = time.perf_counter()
        for _ in range(32):
            loop.frame(pass_mask)
        renderer.synchronize()
        pipelined_frame_ms = (time.perf_counter() - t_pipe) / 32 * 1e3
        frames_rendered += 44

    # ---- untimed priming, independent of --warmup: the first multi-frame call makes the library create its
    # frames-in-flight slots (streams, ~1 GB of path state each at 1080p; ~14 ms) - with --warmup 0 or 1 that
    # would otherwise land inside the timed region
    loop.frames(16, pass_mask)  # one full wavefront at 1080p: every launch of a run has the size of the timed ones
    # ---- warmup
    loop.frames(args.warmup, pass_mask)
    frames_rendered += 16 + args.warmup + args.steps
    if use_dist:
        compose()  # warm the composition path too (RCCL connects its send / recv channels on first use, the library allocates its tile buffers)
    loop.reset()
    renderer.reset_stats()
    renderer.set_option("time_kernels", 1)

    # ---- timed region: exactly K steps (+ the composition gather for N > 1)
    sync_all()
    t0 = time.perf_counter()
    loop.frames(args.steps, pass_mask)  # static camera: the library may batch / overlap frames (uh_render_frames)
    if os.environ.get("UH_BENCH_DEBUG"):
        renderer.synchronize()
        print(f"[debug] frames done at {time.perf_counter() - t0:.4f} s", file=sys.stderr)
    if use_dist:
        compose()
    if os.environ.get("UH_BENCH_DEBUG"):
        renderer.synchronize()
        print(f"[debug] composition done at {time.perf_counter() - t0:.4f} s", file=sys.stderr)
    sync_all()
    elapsed = time.perf_counter() - t0

    st = renderer.get_stats()
    my_rays = float(st.path_rays)
    my_closest = float(st.rays[rr.RAY_BOUNCE] + (st.camera_tree_rays if st.camera_grid_cells else st.rays[rr.RAY_PRIMARY]))  # the rays k_trace_closest traced

    # ---- the same K steps once more with the sun grid and the camera grid off (every ray walks the tree): the figure that owes
    # nothing to a structure built outside the timed region. Same protocol: priming wavefront, barrier, K frames (+ composition), barrier.
    renderer.set_option("time_kernels", 0)
    camera_grid = {
        "in_use": bool(st.camera_grid_cells),
        "build_ms": st.camera_grid_build_ms,       # once per camera at rest (and geometry), on the device, OUTSIDE the timed region
        "pixels": st.camera_grid_cells,
        "entries": st.camera_grid_entries,
        "bytes": st.camera_grid_bytes,
        "mean_list": st.camera_grid_mean_list,
        "handed_to_tree": st.camera_tree_rays / max(st.rays[rr.RAY_PRIMARY], 1),  # share of the primary rays whose pixel lists too many packets
        "tests_per_ray": camera_tests_per_ray,     # triangle tests per primary ray served by the grid (no node visit)
    }
    elapsed_tree, tree_rays = None, 0.0
    if not args.no_tree_walk:
        renderer.set_option("sun_grid", 0)
        renderer.set_option("camera_grid", 0)
        loop.frames(16, pass_mask)
        loop.reset()
        renderer.reset_stats()
        sync_all()
        t1 = time.perf_counter()
        loop.frames(args.steps, pass_mask)
        if use_dist:
            compose()
        sync_all()
        elapsed_tree = time.perf_counter() - t1
        tree_rays = float(renderer.get_stats().path_rays)
        renderer.set_option("sun_grid", 1)
        renderer.set_option("camera_grid", 1)
        frames_rendered += 16 + args.steps
    # ---- and once more with the sun grid's lists kept a second time as 64-byte records that carry their packet (option sun_grid_inline_max_mb
    # raised: round 4's default - 1.4 GB for this scene; since round 5 the default keeps those records within four times the packet array)
    inline_records = None
    if not args.no_tree_walk and sun_grid["in_use"] and not use_dist:
        renderer.set_option("sun_grid_inline_max_mb", 16384)
        loop.frames(16, pass_mask)
        loop.reset()
        renderer.reset_stats()
        sync_all()
        t2 = time.perf_counter()
        loop.frames(args.steps, pass_mask)
        sync_all()
        e2 = time.perf_counter() - t2
        s2 = renderer.get_stats()
        inline_records = {"value": float(s2.path_rays) / e2 / 1e6, "ms_per_step": e2 / args.steps * 1e3, "sun_grid_bytes": int(s2.sun_grid_bytes),
                          "sun_grid_bytes_over_packet_array": s2.sun_grid_bytes / max(64.0 * scene.num_triangles, 1.0)}
        renderer.set_option("sun_grid_inline_max_mb", -1)
        frames_rendered += 16 + args.steps
    if use_dist:
        t = [elapsed, my_rays, my_closest, st.trace_closest_ms, elapsed_tree or 0.0, tree_rays]
        tmax, tsum = rdzv.allreduce(t, "max"), rdzv.allreduce(t, "sum")
        elapsed, total_rays = tmax[0], tsum[1]
        elapsed_tree, tree_rays = (tmax[4] if elapsed_tree is not None else None), tsum[5]
    else:
        total_rays = my_rays

    if rank == 0:
        out = {
            "metric": f"Mrays/s (path rays: primary + bounce + sun-shadow + light-shadow) at {W}x{H}, {args.steps * args.spp} spp = {args.steps} frames x {args.spp} spp; "
                      "camera and sun at rest, the one-off builds of the camera grid and the sun grid outside the timed region (value_with_grid_builds charges them to these frames, value_tree_walk uses neither grid)",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            # ranks of the library's own RCCL communicator (ncclCommCount of what uh_rccl_attach made: the tile gather of every config
            # and config 2's reservoir all-gather run over it); null = no RCCL in this run (single process)
            "rccl_library_comm_ranks": (renderer.rccl_comm_count() or None) if use_dist else None,
            "hip": dict(zip(("built_with", "runtime"), rr.hip_versions()), runtime_mismatch=(rr.load_library().uh_last_error(None) or b"").decode() or None),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            # the same steps with options sun_grid = 0 and camera_grid = 0 (every ray walks the tree), and with the two grids' one-off
            # builds charged to THIS run's K frames: they are built once per (geometry, sun direction) / (geometry, camera), on the
            # device, before the timed region
            "value_tree_walk": tree_rays / elapsed_tree / 1e6 if elapsed_tree else None,
            "ms_per_step_tree_walk": elapsed_tree / args.steps * 1e3 if elapsed_tree else None,
            "value_with_grid_builds": total_rays / (elapsed + ((sun_grid["build_ms"] if sun_grid["in_use"] else 0.0) + (camera_grid["build_ms"] if camera_grid["in_use"] else 0.0)) * 1e-3) / 1e6,
            "value_sun_inline_records": inline_records,  # the same steps with the sun grid's 64-byte records beyond the default memory budget
            "sun_grid": sun_grid,
            "camera_grid": camera_grid,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE.json configs[{args.config}]: {scene.name} synthetic scene ({scene.num_triangles} tris, {scene.num_meshes} meshes, textured materials), "
                f"{W}x{H}, {args.spp} spp/frame x {args.steps} frames, 5 bounces, sky + sun shadow rays"
                + (f", {len(scene.lights)} lights " + ("ReSTIR DI" if view.use_ris_light_sampling else "uniform sampling") if view.lights_enabled else ", lights off"),
                "rays_per_frame": total_rays / args.steps,
                "partition": f"{args.tile}x{args.tile} tiles round-robin over {world} rank(s), 1 RCCL gather inside the library (uh_rccl_gather_tiles)" if world > 1 else "single GPU",
                "frame_by_frame_ms": frame_by_frame_ms,      # fully serial: one stream, one frame, per-kernel event timing on
                "interactive_frame_ms": interactive_frame_ms,  # default options, a synchronisation after every frame
                "pipelined_frame_ms": pipelined_frame_ms,      # one uh_render_frame per frame, four frames in flight, no batching: a moving camera
                "serial_kernel_ms_per_frame": serial_ms_per_frame,  # HIP-event time by kernel kind, one 16-frame wavefront alone on the GPU, nothing overlapped
            },
            # the kernel the serialised wavefront spends most of its traversal time in: the closest-hit walk of the bounce rays - or, with
            # many lights (config 2), the light shadow rays' walk
            "roofline": (roofline(args, st, alone_light_ms, alone_light_rays, float(st.rays[rr.RAY_LIGHT_SHADOW]), light_nodes_per_ray, light_tris_per_ray, elapsed, signature(args, scene, W, H),
                                  rays_per_frame=total_rays / args.steps, kernel="k_trace_shadow_light", launches=st.trace_light_launches, kernel_ms=st.trace_light_ms)
                         if alone_light_ms * max(st.trace_light_launches, 0) > alone_ms * max(st.trace_closest_launches, 1) else
                         roofline(args, st, alone_ms, alone_rays, my_closest, nodes_per_ray, tris_per_ray, elapsed, signature(args, scene, W, H), rays_per_frame=total_rays / args.steps)),
        }
        if os.environ.get("UH_BENCH_SIGNATURE"):  # tools/pmc_summary.py stamps counter profiles with it
            json.dump(dict(signature(args, scene, W, H), frames_total=frames_rendered), open(os.environ["UH_BENCH_SIGNATURE"], "w"))
        parity_failed = False
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            out["cpu_baseline"], ref = cpu_baseline(args, scene)
            if ref is not None:
                # the image this run TIMED, checked: the same scene object, resolution, view and frame numbers, through the same
                # batched uh_render_frames path as the timed region, against the frames the CPU oracle has just rendered
                out["parity"] = parity(args, rr, renderer, scene, pass_mask, W, H, ref)
                parity_failed = not (out["parity"]["max_pixel_l2"] <= 1e-3 and out["parity"]["ray_counts_equal"])
        print(json.dumps(out), flush=True)
        if parity_failed:
            print("[bench] PARITY FAILED: the timed workload's image differs from the oracle's beyond 1e-3 per pixel, or the ray counts differ", file=sys.stderr, flush=True)
            sys.exit(3)
    if use_dist:
        rdzv.barrier()
        renderer.rccl_detach()
        rdzv.close()


def signature(args, scene, W, H):
    """what a committed counter profile must match before its per-launch figures are quoted for this run"""
    return {"config": args.config, "width": W, "height": H, "spp": args.spp, "triangles": int(scene.num_triangles), "opts": sorted(args.opt),
            "emulate_world": args.emulate_world, "gpus": args.gpus}


def load_profile(sig):
    """profiles/bench_counters.json (tools/pmc_summary.py over the rocprofv3 --pmc passes of bench command lines): one profile per
    command line, keyed by its signature; per-launch counter figures of the kernels, used only by the run whose signature equals the profile's"""
    path = os.path.join(ROOT, "profiles", "bench_counters.json")
    try:
        prof = json.load(open(path))
    except Exception:
        return None
    for p in prof.get("profiles", [prof]):
        if p.get("signature") == sig:
            return p
    return None


# bytes a ray walk STREAMS per ray (wide coalesced reads, which gfx950's FETCH_SIZE reports at half): origin and direction quads through
# the LDS pool's DMA; the light rays' walk three quads (origin, throughput, radiance) and its queue entry. Everything else it reads is
# a scattered record read, which the counter reports exactly (tools/microbench/fetch_size.hip).
STREAMED_READ_BYTES_PER_RAY = {"k_trace_closest": 32.0, "k_trace_shadow_light": 52.0}


def classify_bound(k, hbm_frac, hbm_frac_raw):
    """What limits the dominant kernel, read from the counter profile (never a literal): the busiest of the resources the
    passes measured. `hbm` = HBM-side bytes / serialised launch time / 8 TB/s (corrected; raw in the note), `vmem-issue` = the
    CU's texture addresser busy fraction (every <= 16-byte lane load or store takes a slot), `valu` = VALU issue fraction at
    the nominal 2 clk per wave instruction. Returns (bound, note); without a matching profile the bound is unknown."""
    if not k:
        return None, "no counter profile matches this run's configuration: the bound is not asserted"
    cand = {"hbm": hbm_frac, "vmem-issue": k.get("ta_busy_frac"), "valu": k.get("issue_frac")}
    cand = {a: b for a, b in cand.items() if b is not None}
    if not cand:
        return None, "the matching counter profile carries no busy fractions"
    order = sorted(cand, key=cand.get, reverse=True)
    bound = order[0]
    # two resources within 0.1 of each other share the name: the kernel sits on both
    if len(order) > 1 and cand[order[0]] - cand[order[1]] < 0.1:
        bound = f"{order[0]}/{order[1]}"

    def fmt(x):
        return "n/a" if x is None else f"{x:.2f}"

    note = (f"counter passes (profiles/bench_counters.json): texture addresser busy {fmt(k.get('ta_busy_frac'))}, texture data unit busy {fmt(k.get('td_busy_frac'))}, "
            f"VALU issue {fmt(k.get('issue_frac'))} at 2 clk per wave instruction, lane utilisation {fmt(k.get('lane_utilisation'))}, "
            f"HBM-side bytes {fmt(hbm_frac)} of the 8 TB/s peak ({fmt(hbm_frac_raw)} as the counters report them): the busiest resource names the bound")
    return bound, note


def roofline(args, st, alone_ms, alone_rays, my_closest, nodes_per_ray, tris_per_ray, elapsed, sig, rays_per_frame=None, prof=False, kernel="k_trace_closest", launches=None, kernel_ms=None):
    """Roofline block of the dominant kernel (`kernel`: k_trace_closest, or k_trace_shadow_light - the light shadow rays' walk - with its
    launch count and summed HIP-event time in `launches` / `kernel_ms`; my_closest / alone_rays are then that kernel's rays) - every figure
    physical and <= 1 by construction.
    traffic   HBM-side bytes per launch from the rocprofv3 counter passes of this command line (profiles/bench_counters.json),
              scaled by rays per launch - quoted only when the profile's signature equals this run's. Correction as
              MI355X_MICROARCH.md prescribes AND calibrates: gfx950's FETCH_SIZE reports half of a wide streaming read (16 B per
              lane, coalesced) and - measured here on byte counts known by construction, tools/microbench/fetch_size.hip,
              profiles/r05_fetch_size_calibration.txt - exactly the 64-byte lines of scattered record reads (64-, 48-, 16- and
              4-byte gathers alike). A ray walk streams 32 (light rays: 52) bytes per ray and gathers everything else, so
              traffic = FETCH_SIZE + WRITE_SIZE + half the streamed bytes. `all_reads_doubled` keeps round 2-4's figure
              (2 x FETCH_SIZE + WRITE_SIZE: right for a kernel that only streams, an upper bound here), `uncorrected` the counters' own
    achieved  traffic / the kernel's SERIALISED launch duration, measured live with HIP events on the library's stream with
              the kernel alone on the GPU in launches of the timed size (frac = achieved / 8 TB/s)
    overlapped  the same bytes over the timed region's average launch duration, during which other frames' kernels share the chip
    frame_hbm_frac  HBM-side bytes of ALL kernels PER FRAME (counter passes, summed over every dispatch, divided by the frames the
              profiled run rendered) / ms_per_step / 8 TB/s. Per frame: it does not depend on how many frames a launch carries
              (--steps 20 = a 16-frame and a 4-frame wavefront); it is scaled only by rays per frame when the profile records its own
    algorithmic  SURVEY 8d's work metric at its contract price (128 B per node visit) and at the real record size (48 B): bytes the
              caches serve, not a roofline fraction
    bound     the busiest resource in the counter profile (classify_bound), not a literal
    prof      the counter profile (tests hand one in); False = profiles/bench_counters.json when its signature matches"""
    n_launches = st.trace_closest_launches if launches is None else launches
    launches = max(n_launches, 1)
    avg_ms = (st.trace_closest_ms if kernel_ms is None else kernel_ms) / launches
    rays_per_launch = my_closest / launches
    per_ray_128 = 48.0 + nodes_per_ray * 128.0 + tris_per_ray * 48.0
    per_ray_48 = 48.0 + nodes_per_ray * 48.0 + tris_per_ray * 48.0    # nodes are 48-byte records (csrc/bvh.h)
    if prof is False:
        prof = load_profile(sig)
    k = (prof or {}).get("kernels", {}).get(kernel, {})
    traffic_x2 = k.get("hbm_bytes_per_launch")  # 2 x FETCH_SIZE + WRITE_SIZE
    prof_rays = k.get("rays_per_launch") or k.get("closest_rays_per_launch")
    if traffic_x2 and prof_rays:
        # the profiled run's launches may carry another number of frames than this run's: a LAUNCH's traffic goes with its rays
        traffic_x2 *= rays_per_launch / prof_rays
    # what the counters report (FETCH_SIZE + WRITE_SIZE) and the calibrated figure: only the streamed reads are reported at half
    raw_share = (k.get("hbm_bytes_per_launch_uncorrected") or 0.0) / k["hbm_bytes_per_launch"] if traffic_x2 else None
    streamed = STREAMED_READ_BYTES_PER_RAY.get(kernel, 0.0) * rays_per_launch
    traffic = min(traffic_x2, traffic_x2 * raw_share + 0.5 * streamed) if (traffic_x2 and raw_share) else traffic_x2
    # serialised duration of a launch of the timed size: the calibration launches are the same wavefronts, alone
    serial_ms = alone_ms * (rays_per_launch / alone_rays) if (alone_ms > 0 and alone_rays > 0) else None
    achieved = traffic / (serial_ms * 1e-3) / 1e9 if (traffic and serial_ms) else None
    achieved_x2 = traffic_x2 / (serial_ms * 1e-3) / 1e9 if (traffic_x2 and serial_ms) else None
    overlapped = traffic / (avg_ms * 1e-3) / 1e9 if (traffic and avg_ms > 0) else None
    frame_bytes = (prof or {}).get("frame_hbm_bytes")
    frame_bytes_lo = (prof or {}).get("frame_hbm_bytes_uncorrected")
    prof_rpf = (prof or {}).get("rays_per_frame")
    if frame_bytes and prof_rpf and rays_per_frame:
        # a FRAME's bytes go with the rays of a frame (same scene and camera by the signature: a factor of 1 up to the frame numbers)
        frame_bytes *= rays_per_frame / prof_rpf
        frame_bytes_lo = frame_bytes_lo * rays_per_frame / prof_rpf if frame_bytes_lo else None
    ms_per_step = elapsed / args.steps * 1e3
    uncorrected = raw_share  # share of the doubled figure that the raw counters report
    # vector-memory issue: every per-lane load or store of <= 16 B takes one slot of the CU's texture addresser / data path, and
    # that path retires about one lane per clock (profiles/r02_microbench_rates.txt). Lane operations of one closest-hit ray: 3
    # per node visit (48-B node), 3 per triangle tested (48-B packet), 3 for the ray (2 LDS-DMA, by queue position) + hit.
    lane_ops = 3.0 * nodes_per_ray + 3.0 * tris_per_ray + 3.0
    peak_lane_rate = 256 * 2.4e9  # CUs x max clock (MI355X_MICROARCH.md); the clock under load is lower
    alone_rate = alone_rays * lane_ops / (alone_ms * 1e-3) if alone_ms > 0 else 0.0
    ref = serial_ms or avg_ms
    frac = (achieved / HBM_PEAK_GBS) if achieved is not None else None
    # the bound is classified on the profile's own serialised launch (bytes and duration from the same counter pass)
    prof_frac_raw = (k["hbm_bytes_per_launch_uncorrected"] / (k["launch_ns"] * 1e-9) / (HBM_PEAK_GBS * 1e9)) if (k.get("hbm_bytes_per_launch_uncorrected") and k.get("launch_ns")) else None
    prof_frac = frac
    if k.get("hbm_bytes_per_launch") and k.get("launch_ns"):
        prof_bytes = k["hbm_bytes_per_launch"]
        if k.get("hbm_bytes_per_launch_uncorrected") and prof_rays:
            prof_bytes = min(prof_bytes, k["hbm_bytes_per_launch_uncorrected"] + 0.5 * STREAMED_READ_BYTES_PER_RAY.get(kernel, 0.0) * prof_rays)
        prof_frac = prof_bytes / (k["launch_ns"] * 1e-9) / (HBM_PEAK_GBS * 1e9)
    bound, bound_note = classify_bound(k, prof_frac, prof_frac_raw)
    r = {
        "kernel": kernel,
        "bound": bound,
        "bound_note": bound_note,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": frac,
        "traffic": traffic,
        "traffic_source": ((prof or {}).get("source", "") + " - signature-matched to this command line, scaled by rays per launch; not collected in this run") if traffic
        else "no committed counter profile matches this run's configuration: traffic is null rather than borrowed",
        "serial_launch_ms": serial_ms,
        "overlapped": {"avg_launch_ms": avg_ms, "achieved": overlapped, "frac": (overlapped / HBM_PEAK_GBS) if overlapped else None,
                       "note": "timed region: launches of up to four wavefronts overlap on two streams each, so a launch's duration includes time the chip gave to other kernels"},
        "frame_hbm_frac": (frame_bytes / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9)) if frame_bytes else None,
        "frame_hbm_bytes": frame_bytes,
        # the x2 FETCH_SIZE correction is calibrated for coalesced wide reads; these kernels gather 16-B lanes from 48/64-byte
        # records, for which it may overstate by up to 2x. The same fractions from the counters AS REPORTED are the lower bounds.
        "uncorrected": {"frac": (achieved_x2 / HBM_PEAK_GBS * uncorrected) if (achieved_x2 is not None and uncorrected) else None,
                        "frame_hbm_frac": (frame_bytes_lo / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9)) if frame_bytes_lo else None},
        # round 2-4's figure: every read doubled (right for a kernel that only streams; an upper bound for one that gathers records)
        "all_reads_doubled": {"traffic": traffic_x2, "achieved": achieved_x2, "frac": (achieved_x2 / HBM_PEAK_GBS) if achieved_x2 is not None else None},
        "streamed_read_bytes_per_launch": streamed,
        "launches": n_launches,
        "rays_per_launch": rays_per_launch,
        "nodes_per_ray": nodes_per_ray,
        "tris_per_ray": tris_per_ray,
        "algorithmic": {
            "bytes_per_ray_at_128B_nodes": per_ray_128,
            "gbps_at_128B_nodes": rays_per_launch * per_ray_128 / (ref * 1e-3) / 1e9 if ref else None,
            "bytes_per_ray_at_48B_nodes": per_ray_48,
            "gbps_at_48B_nodes": rays_per_launch * per_ray_48 / (ref * 1e-3) / 1e9 if ref else None,
            "over": "serial_launch_ms" if serial_ms else "overlapped avg_launch_ms",
            "note": "SURVEY 8d work metric: bytes REQUESTED per ray; L1 / L2 / Infinity Cache serve most of them, so this may exceed the HBM peak - not a roofline fraction",
        },
        "limiter": {
            "what": "the CU's vector-memory pipeline: about one <=16-byte lane operation per clock per CU",
            "lane_ops_per_ray": lane_ops,
            "unit": "G lane-ops/s",
            "peak": peak_lane_rate / 1e9,
            "achieved_alone": alone_rate / 1e9,
            "frac_alone": alone_rate / peak_lane_rate,
        },
        "valu": {kk: k.get(kk) for kk in ("wave_instr_per_launch", "issue_frac", "lane_utilisation", "ta_busy_frac", "td_busy_frac")} if k else None,
        # every kernel against the HBM roofline, alone on the GPU (counter passes serialise the kernels): HBM-side bytes per
        # launch over the launch's duration from the counter CSV's own timestamps
        # [every read doubled, as reported]: the first is right for the kernels that stream (k_generate, k_finish_sample, most of
        # k_shade_hit's state), the second for those that gather records (the walks, the grids) - tools/microbench/fetch_size.hip
        "hbm_frac_by_kernel": ({name: [round(v["hbm_bytes_per_launch"] / (v["launch_ns"] * 1e-9) / 8e12, 3),
                                       round((v.get("hbm_bytes_per_launch_uncorrected") or 0.0) / (v["launch_ns"] * 1e-9) / 8e12, 3)]
                                for name, v in prof["kernels"].items() if name.startswith("k_") and v.get("launch_ns") and v.get("hbm_bytes_per_launch")}
                               if prof else None),
        "trace_closest_ms": st.trace_closest_ms,
        "trace_shadow_ms": st.trace_shadow_ms,
        "trace_light_ms": getattr(st, "trace_light_ms", None),
        "shade_ms": st.shade_ms,
    }
    return r


def parity(args, rr, renderer, scene, pass_mask, W, H, ref):
    """render the frames the CPU oracle rendered, from the same initial state, and compare linear radiance per pixel"""
    import numpy as np

    frames, ref_acc, ref_rays = ref
    renderer.synchronize()
    renderer.reset_accumulation()
    if pass_mask & rr.PASS_RESTIR:
        zero = np.zeros((H, W), dtype=rr.types.RESERVOIR_DTYPE)
        for which in (0, 1, 2):
            renderer.write_reservoirs(which, zero)  # the oracle starts from zeroed reservoir history
    renderer.reset_stats()
    view = scene.make_view(W, H, samples_per_frame=args.spp) if args.spp != 1 else scene.make_view(W, H)
    loop = rr.FrameLoop(renderer, view)
    loop.frames(frames, pass_mask)
    acc = renderer.read_accumulation()[..., :3].astype(np.float64) / float(frames * args.spp)
    want = ref_acc[..., :3].astype(np.float64) / float(frames * args.spp)
    d2 = np.sum((acc - want) ** 2, axis=-1)
    s = renderer.get_stats()
    return {
        "frames": frames,
        "pixels": int(W * H),
        "max_pixel_l2": float(np.sqrt(d2.max())),
        "image_rms": float(np.sqrt(d2.mean())),
        "pixels_bit_identical": int(np.sum(d2 == 0.0)),
        "ray_counts_equal": bool(s.path_rays == ref_rays),
        "rays": int(s.path_rays),
        "tolerance": 1e-3,
        "against": "oracle/oracle.cpp frames of the cpu_baseline leg (same scene object, tex size, resolution, flags, frame numbers)",
    }


def usable_cores():
    """host threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(args, scene):
    """The CPU oracle (port of the reference shaders, oracle/oracle.cpp) on this box's host cores,
    same scene / camera / flags, on a bounded sample of the workload."""
    import oracle_api as oa
    import rust_renderer_amd as rr

    native = oa.build_native()  # -O3 -march=native for THIS box's cores (the portable .so is -O3 -mavx2 -mfma)
    w, h, frames = (int(x) for x in args.cpu_sample.split("x"))
    cores = usable_cores()
    o = scene.upload(oa.OracleRenderer(w, h, threads=cores))
    loop = rr.FrameLoop(o, scene.make_view(w, h, samples_per_frame=args.spp) if args.spp != 1 else scene.make_view(w, h))
    t0 = time.perf_counter()
    done = 0
    while done < frames and (done == 0 or time.perf_counter() - t0 < args.cpu_seconds):
        loop.frame(rr.PASS_ALL if args.config == 2 else rr.PASS_REFERENCE_PT)
        done += 1
    frames = done
    dt = time.perf_counter() - t0
    s = o.get_stats()
    ref = (frames, o.read_accumulation(), int(s.path_rays)) if (w, h) == (args.width, args.height) else None
    return {
        "value": s.path_rays / dt / 1e6,
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "sample": f"same scene and camera at {w}x{h}, {frames} frames x 1 spp, 5 bounces ({s.path_rays} rays in {dt:.1f} s)",
        "build": "g++ -O3 -march=native -ffp-contract=off" if native else "g++ -O3 -mavx2 -mfma -ffp-contract=off (no compiler on this box for a native build)",
        "acceleration_structure": "the oracle's own median-split BVH2 (the parity checker, not a tuned CPU tracer): a reported baseline, not a speed-up claim",
    }, ref


if __name__ == "__main__":
    main()
