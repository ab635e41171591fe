#!/usr/bin/env python3
"""bench.py - Mrays/s of the Sol-R per-pixel rendering path on MI355X.

Workload (BASELINE.json configs[1]): Cornell box, 1920x1080, 3 reflection bounces +
shadow rays, synthetic scene from sol-r_amd/scenes.py.  One "step" = one frame:
k_standardRenderer over this rank's row strip (scene resident in HBM, uploaded before
the timed region) and, for N > 1, the RCCL gather of the RGB strips to rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = (closest-hit walks + shadow walks of the whole
frame) * steps / wall time / 1e6, summed over all ranks, max wall time over ranks.
`roofline` prices the renderer kernel against HBM bandwidth with the algorithmic bytes
of DESIGN.md (67 B per pixel + one read of the scene), timed with HIP events on the
launch stream.  `cpu_baseline` times the CPU oracle (a port of the reference algorithm,
see oracle/solr_oracle.h) on the host cores of the same box on a bounded sample.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
PREROLL_FRAMES = 48    # untimed setup frames before the W warmup steps
BYTES_PER_PIXEL = 67    # 16 (ids write) + 32 (float framebuffer write) + 16 (ids read, refinement passes) + 3 (RGB)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--scene", default="cornell", choices=["cornell", "height_field", "molecule", "irt_model", "obj_model", "swc_morphology", "pdb_molecule"])
    ap.add_argument("--scene-file", default="",
                    help="--scene irt_model / obj_model: the file (default: the reference's samples under tests/golden)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--graphics-level", type=int, default=4, help="experiments only; the metric is quoted at 4 (glFull)")
    ap.add_argument("--tile-scheduling", type=int, default=1, help="0 raster order, 1 automatic (default), 2 cost order")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="consecutive frames rotate over this many streams and buffer sets of the engine, so a "
                         "frame's tail overlaps the next frame's start; 1: the reference's one frame at a time. "
                         "Default: 2 at N = 1, 3 at N > 1 (strips are one round of waves with a long tail)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def main():
    args = parse()
    if os.environ.get("SOLR_BENCH_DEBUG"):
        import faulthandler
        import signal
        faulthandler.register(signal.SIGUSR1, all_threads=True)  # `timeout -s USR1` prints where it hangs
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world),
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    torch = dist = None
    # SOLR_BENCH_FORCE_DIST=1 runs the N > 1 code path (process group, strip binding, gather) with a
    # single rank: a 1-GPU box can then exercise it against RCCL
    distributed = world > 1 or os.environ.get("SOLR_BENCH_FORCE_DIST") == "1"
    if distributed:
        # torch BEFORE the engine library: torch brings its own copy of the HIP runtime and importing it
        # into a process in which another copy is already initialised hangs
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    solr = importlib.import_module("sol-r_amd")
    hip = solr.hip_lib()
    if hip.solr_hip_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")

    W, H = args.width, args.height
    first_row, nb_rows, rows_per_rank = solr.strip_rows(rank, world, H)

    # ---- scene (built once on every rank: the scene is replicated, pixels are sharded)
    k = solr.Kernel(engine="hip", device=local_rank)
    builder = getattr(solr.scenes, args.scene)
    kw = dict(width=W, height=H, iterations=args.iterations)
    if args.scene not in ("cornell", "irt_model", "obj_model", "swc_morphology", "pdb_molecule"):
        kw.pop("iterations")
    if args.scene in ("irt_model", "obj_model", "swc_morphology", "pdb_molecule"):
        default = {"irt_model": "model_subset.irt", "obj_model": "cornell.obj", "swc_morphology": "pyramidal.swc",
                   "pdb_molecule": "1BNA.pdb"}[args.scene]
        path = args.scene_file or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", default)
        builder(k, path, **kw)
    else:
        builder(k, **kw)
    if args.graphics_level != 4:
        k.set_scene_info(graphicsLevel=args.graphics_level)
    hip.solr_hip_set_variant(args.variant)
    hip.solr_hip_set_tile_scheduling(args.tile_scheduling)
    if args.frames_in_flight <= 0:
        args.frames_in_flight = 3 if distributed else 2
    if not distributed:
        hip.solr_hip_set_frames_in_flight(args.frames_in_flight)
    pipe = None
    if distributed:
        # every frame: render on one of the engine's streams, then the gather in order on that same
        # stream, while the next frames render on the other streams (StripPipeline)
        pipe = solr.StripPipeline(dist, torch, hip, W, H, rank, world, local_rank=local_rank,
                                  frames_in_flight=args.frames_in_flight)

    # first frame through the full host protocol: uploads scene, materials, randoms
    k.L.SolRx_Render(0.0)
    k.check(0, "first frame")
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731

    def render():
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))

    def frame():
        # N = 1: the renderer alone.  N > 1: the renderer writes RGB8 straight into a strip buffer and the
        # single collective of the path - strips -> rank 0, RCCL over xGMI - follows (StripPipeline)
        if pipe is None:
            render()
        else:
            pipe.frame(render)

    def sync():
        if distributed:
            pipe.drain()
        else:
            hip.solr_hip_synchronize()

    def barrier():
        if distributed:
            dist.barrier()

    # ---- ray census of this rank's strip (untimed; input-determined)
    counts = (C.c_ulonglong * 8)()
    hip.solr_hip_render_counting(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles),
                                 counts)
    k.check(0, "ray census")
    rays_local = int(counts[0]) + int(counts[1])

    # a frame loop that does not come back is reported, not sat out: the multi-GPU loop chains work
    # across streams and ranks, and a stuck collective would otherwise hang the whole launch
    import threading

    def stuck():
        print("bench.py: rank %d made no progress for 600 s (frames in flight %d, distributed %s): giving up"
              % (rank, args.frames_in_flight, distributed), file=sys.stderr, flush=True)
        os._exit(3)

    watchdog = threading.Timer(600.0, stuck)
    watchdog.daemon = True
    watchdog.start()

    # setup, untimed: let the clocks and the tile-cost feedback of the engine settle before the W
    # warmup steps (a frame is 0.4 ms; W = 3 alone is 1.2 ms of GPU work, shorter than the power ramp)
    for _ in range(PREROLL_FRAMES):
        frame()
    sync()
    for _ in range(args.warmup):
        frame()
    sync()
    hip.solr_hip_kernel_time(None, 1)
    # event pairs cost launch gaps: every launch at N = 1, every fourth when frames are short (N > 1)
    hip.solr_hip_enable_timing(4 if distributed else 1)
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    t_issued = time.perf_counter()
    sync()
    barrier()
    t1 = time.perf_counter()
    hip.solr_hip_enable_timing(0)
    watchdog.cancel()
    k.check(0, "timed frames")
    elapsed = t1 - t0
    launches = C.c_int(0)
    kernel_ms = hip.solr_hip_kernel_time(C.byref(launches), 1)
    kernel_basis = "HIP events around every launch of the timed region"
    if distributed and args.frames_in_flight > 1:
        kernel_basis = ("HIP events around every fourth launch of the timed region; with %d frames in flight the launches "
                        "overlap, so this is the latency of a launch, not its share of the GPU" % args.frames_in_flight)
    if not distributed and hip.solr_hip_get_frames_in_flight() > 1:
        # with two frames in flight consecutive launches overlap and an event pair around one of them
        # spans parts of two frames; the kernel's own duration is taken from a short one-at-a-time
        # segment after the timed region (same frame, same buffers)
        hip.solr_hip_set_frames_in_flight(1)
        for _ in range(8):
            frame()
        sync()
        hip.solr_hip_enable_timing(1)
        for _ in range(16):
            frame()
        sync()
        hip.solr_hip_enable_timing(0)
        kernel_ms = hip.solr_hip_kernel_time(C.byref(launches), 1)
        hip.solr_hip_set_frames_in_flight(args.frames_in_flight)
        kernel_basis = "HIP events around each of 16 launches issued one at a time after the timed region"

    rays_total = rays_local
    if distributed:
        t = torch.tensor([elapsed, float(rays_local), kernel_ms / max(launches.value, 1)], dtype=torch.float64,
                         device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays_total = int(tsum[1])
        kernel_avg_ms = float(tmax[2])
        # every rank empties its C stdout buffer (RCCL's version banner sits there until exit) before rank 0
        # may print: the JSON line is then the last thing the job writes to stdout
        C.CDLL(None).fflush(None)
        sys.stdout.flush()
        dist.barrier()
    else:
        kernel_avg_ms = kernel_ms / max(launches.value, 1)

    if rank != 0:
        dist.destroy_process_group()
        C.CDLL(None).fflush(None)
        return

    mrays = rays_total * args.steps / elapsed / 1e6
    scene_bytes = (48 * len(flat.boxes) + 128 * len(flat.primitives) + 48 * len(flat.lights) +
                   176 * len(set(int(m) for m in flat.primitives["materialId"])))
    algo_bytes = nb_rows * W * BYTES_PER_PIXEL + scene_bytes  # per launch, rank 0's strip
    achieved = algo_bytes / (kernel_avg_ms * 1e-3) / 1e9 if kernel_avg_ms > 0 else 0.0
    traffic, traffic_source = measured_traffic(args, world)
    valu = measured_valu(args, world)
    out = {
        "metric": "Mrays/s @1920x1080, 3-bounce Cornell",
        "value": round(mrays, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if args.scene not in ("irt_model", "obj_model", "swc_morphology", "pdb_molecule") else "the reference's sample scene file",
        "config": {"workload": "%s %dx%d, %d bounces + shadow rays, %d boxes / %d primitives, row strips of %d" %
                   (args.scene, W, H, si.nbRayIterations, len(flat.boxes), len(flat.primitives), rows_per_rank),
                   "rays_per_frame": rays_total, "closest_hit_walks_rank0": int(counts[0]),
                   "shadow_walks_rank0": int(counts[1]), "lane_nodes": int(counts[2]), "lane_prim_tests": int(counts[3]),
                   "wave_nodes": int(counts[4]), "wave_prim_tests": int(counts[5]), "wave_walks": int(counts[6]) + int(counts[7]), "mpixels_per_s": round(W * H * args.steps / elapsed / 1e6, 2),
                   "host_issue_ms_per_step_rank0": round((t_issued - t0) / args.steps * 1e3, 4),
                   "cost_ordered_launch_rank0": bool(hip.solr_hip_tile_scheduling_active()),
                   "frames_in_flight": int(hip.solr_hip_get_frames_in_flight()),
                   "parallelism": "tile%d" % world},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "k_standardRenderer", "kernel_ms": round(kernel_avg_ms, 5), "kernel_ms_basis": kernel_basis,
                     "algorithmic_bytes": algo_bytes},
    }
    if valu and kernel_avg_ms > 0:
        # what actually bounds this kernel: a wave64 vector instruction occupies one of a
        # CU's four SIMD16 units for four cycles -> 256 CUs x 4 SIMDs x 2.4 GHz / 4 wave-instructions per second
        peak = 256 * 4 * 2.4e9 / 4
        out["roofline"]["valu_issue"] = {
            "insts_per_launch": valu[0], "source": valu[1], "peak_wave_insts_per_s": peak,
            "frac": round(valu[0] / (kernel_avg_ms * 1e-3) / peak, 4),
            "note": "informative: the contract's roofline is the HBM one above; this kernel is vector-issue bound"}

    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(flat, si, ppi, eye, direction, angles, args.cpu_seconds)
    if distributed:
        # RCCL writes its version banner to the C library's stdout buffer when the communicator comes up; on a
        # pipe that buffer is only flushed at exit, after our line.  Tear the group down and flush it first:
        # the JSON line is the last thing on stdout
        dist.destroy_process_group()
    C.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


def measured_traffic(args, world):
    """HBM-side bytes per launch of the renderer from the committed rocprofv3 --pmc passes of this same
    command (profiles/rNN/hbm_traffic.json, written by tools/collect_profiles.py); None when the workload
    differs from the profiled one.  bench.py cannot run the profiler on itself."""
    import glob
    if world != 1 or (args.width, args.height) != (1920, 1080) or args.graphics_level != 4:
        return None, None
    if args.scene == "cornell" and args.iterations != 3:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic.json")), reverse=True):
        try:
            entry = json.load(open(path)).get(args.scene)
        except (OSError, ValueError):
            continue
        if entry:
            return entry["bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def measured_valu(args, world):
    """SQ_INSTS_VALU per launch of the renderer from the committed PMC pass of this same command
    (profiles/rNN/pmc_<scene>.txt), or None."""
    import glob
    if world != 1 or (args.width, args.height) != (1920, 1080) or args.graphics_level != 4:
        return None
    if args.scene == "cornell" and args.iterations != 3:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_%s.txt" % args.scene)), reverse=True):
        try:
            for line in open(path):
                if line.startswith("SQ_INSTS_VALU"):
                    return int(float(line.split()[1])), os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None


def usable_cpus():
    """cores this process may actually use: affinity mask, capped by a cgroup CPU quota if one is set"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(flat, si, ppi, eye, direction, angles, budget_s):
    """The CPU oracle (a port of the reference algorithm) on the host cores of this box, on a bounded
    sample of the same frame.  Thread count: the best of {quota, 2x, 4x quota, all hardware threads} in
    a short calibration (containers here run under a CFS quota far below the hardware thread count, and
    oversubscribing it collapses throughput), then a sustained run of about budget_s seconds."""
    import numpy as np
    from oracle import loader
    L = loader.lib()
    scene = loader.Scene(flat)
    W, H = si.size_x, si.size_y
    pp = np.zeros((H, W, 8), np.float32)
    ids = np.zeros((H, W, 4), np.int32)
    rgb = np.zeros((H, W, 3), np.uint8)
    counts = (C.c_ulonglong * 4)()
    eye, direction, angles = (np.ascontiguousarray(a, np.float32) for a in (eye, direction, angles))

    def one_frame(threads):
        t0 = time.perf_counter()
        L.oracle_render(C.byref(scene.c), C.addressof(si), C.addressof(ppi), eye.ctypes.data, direction.ctypes.data,
                        angles.ctypes.data, 0, H, pp.ctypes.data, ids.ctypes.data, rgb.ctypes.data,
                        C.addressof(counts), threads)
        return time.perf_counter() - t0, int(counts[0]) + int(counts[1])

    hw = len(os.sched_getaffinity(0))
    quota = usable_cpus()
    candidates = sorted(set(min(hw, c) for c in (quota, 2 * quota, 4 * quota, hw)))
    calib = {}
    for c in candidates:
        one_frame(c)
        calib[c] = min(one_frame(c)[0] for _ in range(2))
    threads = min(calib, key=calib.get)
    frames, rays, t0 = 0, 0, time.perf_counter()
    while True:
        _, r = one_frame(threads)
        rays += r
        frames += 1
        if time.perf_counter() - t0 > budget_s or frames >= 2000:
            break
    dt = time.perf_counter() - t0
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "%d full %dx%d frames of the same scene in %.1f s, OpenMP over rows with %d threads "
                      "(affinity %d, CFS quota %d CPUs; calibration s/frame by threads: %s)" %
                      (frames, W, H, dt, threads, hw, quota, {k: round(v, 4) for k, v in calib.items()})}


if __name__ == "__main__":
    main()
