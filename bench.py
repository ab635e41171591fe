#!/usr/bin/env python3
"""bench.py - Mrays/s of the Sol-R per-pixel rendering path on MI355X.

Workload (BASELINE.json configs[1]): Cornell box, 1920x1080, 3 reflection bounces +
shadow rays, synthetic scene from sol-r_amd/scenes.py.  One "step" = one frame DELIVERED:
k_standardRenderer over this rank's row strip (scene resident in HBM, uploaded before
the timed region), for N > 1 the RCCL gather of the RGB strips to rank 0, and the frame's image copied to a
page-locked host image behind it (copy stream; the host takes the image of the frame `frames in flight - 1` back
while the newer ones render - what HipKernel::setFramesInFlight does for the reference's render_begin / render_end).
The primitive ids stay on the device until picking asks (d2h_bitmap).

  python bench.py --gpus 1 --steps 200 --warmup 3
  python bench.py --gpus N ...                      (bare: starts its own N rank processes)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Timing: W warm-up steps, then the timed region - EXACTLY K steps between barrier + synchronisation on both sides, the
last image on the host when it ends, no event or other instrument inside it - is run R >= 25 times back to back (as
many as keep the GPU busy for 0.2 s).  `ms_per_step` and `value` are the MEDIAN region's; `config.step_ms_spread` gives
the fastest and the slowest region, which bracket it by construction.  At N > 1 a region's time is the MAX over ranks
of each rank's own time from the start barrier to its last delivered frame (the end barrier itself is the control
plane's - gloo over TCP - and is reported, not charged).

N > 1 (the launcher, the control plane, the communicator, the delivery routes and the sweep over them live in
bench_dist.py): row strips re-cut by measured cost (solr_hip_balance_strips), gathered to rank 0 with RCCL called from
the engine's C ABI on the stream that rendered the frame - `value` is always a combination that makes that RCCL call
behind every frame, on a communicator of exactly --gpus ranks (exit code 5 otherwise); the reference's equal split is
timed as a second segment, the assembled frame is compared with the frame rank 0 renders alone, per-rank times and the
gather alone are reported, and which communicator mode ran (one for everything, or one per frame in flight).

Rank 0 prints ONE JSON line.  `value` = (closest-hit walks + shadow walks of the whole
frame) * steps / the median region's wall time / 1e6, summed over all ranks, max wall time over ranks.
`roofline` prices the renderer kernel against HBM bandwidth with the algorithmic bytes
of DESIGN.md (51 B per pixel on a first pass + one read of the scene), timed with HIP events on the
launch stream around launches issued one at a time AFTER the timed regions; `config.rates_mrays_per_s` holds, next to
`value`, the frames left in HBM (earlier rounds' headline) and the rates of the reference's own frame
protocol (one frame at a time; cudaRender + d2h_bitmap).  `cpu_baseline` times the CPU oracle (a port of the reference algorithm,
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

import bench_dist  # noqa: E402  (the N > 1 side: launcher, control plane, communicator, delivery route, mode sweep)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
PREROLL_FRAMES = 48    # untimed setup frames before the W warmup steps
# algorithmic bytes per pixel of one launch of the renderer (SURVEY.md section 8d with k_default fused):
FIRST_PASS_BYTES_PER_PIXEL = 51    # pass 0: 16 (ids write) + 32 (float frame buffer write) + 3 (RGB); nothing is read
LATER_PASS_BYTES_PER_PIXEL = 99    # refinement / accumulation passes also read the ids (16) and the frame buffer (32)
SURVEY_FUSED_BYTES_PER_PIXEL = 67  # SURVEY.md 8(d) with k_default fused: + the 16 B ids read of the early-out test
# MI355X_MICROARCH.md: 4 SIMD-32 per CU, a wave64 vector instruction issues over 2 cycles; one scalar unit per CU
VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2
SALU_PEAK_WAVE_INSTS_PER_S = 256 * 2.4e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps per region (default 200 - a 60 ms region, run at least 25 times; --config cfg4: 74, one whole cycle of passes)")
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
                         "Default 3 (strips are one round of waves with a long tail)")
    ap.add_argument("--config", default="", choices=["", "cfg1", "cfg2", "cfg3", "cfg4"],
                    help="a BASELINE.json configuration by name: cfg1 = the default (Cornell 1080p, 3 bounces), cfg2 = "
                         "--scene height_field, cfg3 = --scene molecule, cfg4 = Cornell 3840x2160 through passes 0...73 "
                         "(refinement + 64 accumulated samples, natural depth of field, ambient-occlusion kernel): a "
                         "step is then one pass of that cycle")
    ap.add_argument("--equal-strips", action="store_true",
                    help="N > 1: keep the reference's equal row strips for the headline segment (default: strips re-cut "
                         "by measured cost after the set-up frames, solr_hip_balance_strips; the equal split is then "
                         "timed as a second segment and reported next to it)")
    ap.add_argument("--torch-gather", action="store_true",
                    help="N > 1: gather the strips with torch.distributed's gather (NCCL backend = RCCL) on equal strips "
                         "instead of RCCL called from the engine's C ABI on the frame's own stream (the default)")
    # accepted for the command lines of earlier rounds: both are the default now
    ap.add_argument("--balanced-strips", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--native-gather", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--delivery", default="auto", choices=["auto", "strips", "gathered"],
                    help="N > 1, how the frame reaches the host ('auto', the default: both are timed in short segments, with "
                         "one communicator and with one per frame in flight, and the headline runs on the fastest "
                         "combination whose frame equals the one-GPU frame - config.mode_sweep): 'strips' - every rank copies its strip over its own PCIe "
                         "link into one page-locked image the ranks' processes share (solr_hip_image_share; the reference's "
                         "d2h_bitmap does the same with the devices of its one process) - or 'gathered': rank 0 copies the "
                         "frame the RCCL gather assembled in its HBM (one PCIe link for the whole frame).  The sweep also "
                         "times 'strips' with no RCCL call per frame at all (SOLR_BENCH_COLLECTIVE=0|1 fixes that choice)")
    ap.add_argument("--no-check", action="store_true",
                    help="N > 1: skip the comparison of the gathered frame with the frame rank 0 renders alone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-walk-bound", action="store_true",
                    help="skip the walk's own ceiling (roofline.walk_bound_mrays: a gigabyte of records for one 1080p frame)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()
    if args.config == "cfg2":
        args.scene = "height_field"
    elif args.config == "cfg3":
        args.scene = "molecule"
    elif args.config == "cfg4":
        args.scene, args.width, args.height, args.iterations = "cornell", 3840, 2160, 1
        args.frames_in_flight = 1          # every pass reads what the pass before left in the frame buffers
    if args.steps is None:
        args.steps = 74 if args.config == "cfg4" else 200   # cfg4's passes differ in cost: whole cycles
    if args.frames_in_flight <= 0:
        args.frames_in_flight = 3   # measured: 1 -> 0.311, 2 -> 0.292, 3 -> 0.288, 4 -> 0.286 ms per Cornell frame
    return args


def spread(samples, divide=1.0, window=1):
    """min / median / max of a list of milliseconds (the error bar of a short timed region).  window > 1: of the mean
    over every run of `window` consecutive samples - with frames in flight the launches end in bursts, and only a run
    longer than the flights says what a step takes."""
    xs = [float(x) / divide for x in samples if x is not None and x >= 0.0]
    if window > 1 and len(xs) >= window:
        xs = [sum(xs[i:i + window]) / window for i in range(len(xs) - window + 1)]
    xs.sort()
    if not xs:
        return None
    return {"min": round(xs[0], 5), "median": round(xs[len(xs) // 2], 5), "max": round(xs[-1], 5), "samples": len(xs),
            "window": window}


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    args = parse()
    # bare `python bench.py --gpus N` (no launcher's environment): become the launcher, before anything of the GPU
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(bench_dist.launch(args, __file__))
    cfg4 = args.config == "cfg4"
    if os.environ.get("SOLR_BENCH_DEBUG"):
        import faulthandler
        import signal
        faulthandler.register(signal.SIGUSR1, all_threads=True)  # `timeout -s USR1` prints where it hangs
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    share = os.environ.get("SOLR_BENCH_SHARE_GPU") == "1"   # rehearsal on a one-GPU box: every rank on GPU 0
    if share:
        local_rank = 0

    torch = dist = None
    # SOLR_BENCH_FORCE_DIST=1 runs the N > 1 code path (process group, strips, communicator, gather, check) with a
    # single rank: a 1-GPU box can then exercise it against RCCL
    distributed = world > 1 or os.environ.get("SOLR_BENCH_FORCE_DIST") == "1"
    torch_gather = distributed and args.torch_gather and not cfg4   # (cfg4's ambient-occlusion taps cross the strips:
    native = distributed and not torch_gather                       #  only the library's own communicator trades them)
    balanced = native and not args.equal_strips
    if distributed:
        # torch BEFORE the engine library: torch brings its own copy of the HIP runtime and importing it
        # into a process in which another copy is already initialised hangs
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

    solr = importlib.import_module("sol-r_amd")
    hip = solr.hip_lib()
    visible = hip.solr_hip_device_count()
    if visible < 1 or (local_rank >= visible):
        raise SystemExit("bench.py rank %d needs a GPU (device %d of %d visible): the engine has no CPU fallback"
                         % (rank, local_rank, visible))
    if native:
        dist.init_process_group(backend="gloo")       # control plane only: rendezvous, barrier, timing reduction
    elif distributed:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    W, H = args.width, args.height
    first_row, nb_rows, rows_per_rank = solr.strip_rows(rank, world, H)

    # ---- scene (built once on every rank: the scene is replicated, pixels are sharded)
    k = solr.Kernel(engine="hip", device=local_rank)
    builder = getattr(solr.scenes, args.scene)
    kw = dict(width=W, height=H, iterations=args.iterations)
    if cfg4:
        k.set_post_processing(type=solr.ppe_ambientOcclusion, param1=11000.0, param2=10.0, param3=0)
        kw["maxPathTracingIterations"] = 74
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
    if not distributed or native:
        hip.solr_hip_set_frames_in_flight(args.frames_in_flight)
    pipe = None
    if native:
        # the data path without torch: this rank's strip, K frames in flight on the engine's own streams, and
        # behind every frame the library's own RCCL gather on that frame's stream
        hip.solr_hip_set_strip(first_row, nb_rows)
    elif distributed:
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

    pass_counter = [0]

    def render():
        if cfg4:
            si.pathTracingIteration = pass_counter[0] % 74
            pass_counter[0] += 1
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))

    rccl_ranks = comm_count = None
    delivery_fallback = None
    # what is up right now (bench_dist.Ranks): mode["collective"] - the RCCL gather runs behind every frame (False: the
    # strips meet in the one host image the ranks' processes share and nothing else moves per frame -
    # SOLR_BENCH_COLLECTIVE=0|1 fixes it, else swept; never the headline's)
    R = bench_dist.Ranks(dist, torch, hip, rank, world, collective=os.environ.get("SOLR_BENCH_COLLECTIVE", "1") != "0")
    mode, comm_up, delivery_up = R.mode, R.comm_up, R.delivery_up
    alone = None
    sweep_routes = [args.delivery] if args.delivery != "auto" else ["strips", "gathered"]
    env_per_flight = os.environ.get("SOLR_HIP_COMM_PER_FLIGHT")
    sweep_comms = [env_per_flight[0] != "0"] if env_per_flight else [False, True]
    if args.delivery == "auto":
        args.delivery = "strips"
    if args.delivery == "gathered":
        mode["collective"] = True       # (that route delivers what the gather assembled)
    if native:
        if not cfg4 and not args.no_check:
            # the frame a rank renders ALONE, whole, before there is a communicator: what every assembled frame of the
            # job is compared with (config.mode_sweep).  Every rank renders it - the ranks run the same program, frame
            # for frame: the engine deals its buffer sets (and, with one communicator per frame in flight, its
            # communicators) out by frame number - and rank 0 keeps it
            import numpy as np
            hip.solr_hip_set_strip(0, -1)
            render()
            if rank == 0:
                alone = np.zeros((H, W, 3), np.uint8)
                hip.solr_hip_d2h(C.byref(si), C.c_void_p(alone.ctypes.data), None)
            hip.solr_hip_synchronize()
            k.check(0, "the frame on one GPU")
            hip.solr_hip_set_strip(first_row, nb_rows)
        if not comm_up(None if env_per_flight else sweep_comms[0]):
            raise SystemExit("bench.py rank %d: no communicator" % rank)
        rccl_ranks = int(hip.solr_hip_comm_ranks())
        comm_count = int(hip.solr_hip_comm_count())
        args.delivery, delivery_fallback = delivery_up(args.delivery)

    def frame():
        # N = 1: the renderer alone.  N > 1: the renderer writes RGB8 straight into a strip buffer and the
        # single collective of the path - strips -> rank 0, RCCL over xGMI - follows (the library's own gather on
        # the frame's stream, or StripPipeline)
        if native:
            render()
            if mode["collective"] and hip.solr_hip_gather_strips(0) != 0:
                k.check(-1, "solr_hip_gather_strips")
        elif pipe is None:
            render()
        else:
            pipe.frame(render)

    def sync():
        if pipe is not None:
            pipe.drain()
        else:
            hip.solr_hip_synchronize()

    def barrier():
        if distributed:
            dist.barrier()

    # ---- ray census of this rank's (equal) strip (untimed; input-determined; the frame's total does not depend on
    # how it is cut)
    counts = (C.c_ulonglong * 8)()
    hip.solr_hip_render_counting(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles),
                                 counts)
    k.check(0, "ray census")
    rays_local = int(counts[0]) + int(counts[1])
    rays_first_pass_local = rays_local     # (cfg4: pass 0's, what the oracle's one check frame is compared with)
    if cfg4:
        # the passes differ (refinement passes skip finished pixels and trace deeper, accumulation passes jitter):
        # census of every pass of the cycle, each on the frame buffers the pass before left; a step = a pass,
        # its rays the mean over the cycle
        total = 0
        for it in range(74):
            si.pathTracingIteration = it
            hip.solr_hip_render_counting(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles),
                                         counts)
            total += int(counts[0]) + int(counts[1])
        k.check(0, "ray census of the 74 passes")
        rays_local = total // 74
        si.pathTracingIteration = 0

    # a phase that does not come back is reported, not sat out: the multi-GPU loop chains work across streams and
    # ranks, and a stuck collective would otherwise hang the whole launch.  The timer is re-armed per phase (the
    # whole job may well take longer than one phase's budget: 8 ranks rehearsing on one GPU, cfg4's 74 passes)
    import threading
    phase_budget = float(os.environ.get("SOLR_BENCH_PHASE_TIMEOUT", "600"))
    watch = {"timer": None, "phase": "set-up"}

    def arm(phase):
        if watch["timer"] is not None:
            watch["timer"].cancel()
        watch["phase"] = phase

        def stuck():
            print("bench.py: rank %d: phase '%s' has been running for %.0f s (SOLR_BENCH_PHASE_TIMEOUT; frames in flight "
                  "%d, distributed %s): giving up" % (rank, phase, phase_budget, args.frames_in_flight, distributed),
                  file=sys.stderr, flush=True)
            os._exit(3)

        watch["timer"] = threading.Timer(phase_budget, stuck)
        watch["timer"].daemon = True
        watch["timer"].start()

    def disarm():
        if watch["timer"] is not None:
            watch["timer"].cancel()
            watch["timer"] = None

    arm("set-up frames")

    # ---- the step that is timed: a frame DELIVERED - rendered and, N > 1, gathered to rank 0, and its image on the
    # host (SURVEY.md 8d defines the metric over cudaRender + d2h_bitmap; the north star's path ends at getBitmap).
    # Pipelined as HipKernel::setFramesInFlight does it: the image of a frame is copied to a page-locked host image on
    # a copy stream behind its kernel (behind its gather on rank 0), the host takes the image `lag` frames back while
    # the newer ones render.  The engine runs one buffer set up to a depth of two, two beyond (a third render stream
    # ends up sharing a hardware queue with the copy stream: profiles/r3/readback_probe.txt); the rest is host lag.
    from collections import deque
    # (cfg4: every pass reads the buffers of the pass before - one buffer set - but its image can land while the next
    # pass renders: a set has a second RGB image for that)
    depth = 2 if cfg4 else max(1, args.frames_in_flight)
    engine_sets = 1 if depth <= 2 else 2
    copy_inline = False
    if native and world > 1 and not cfg4 and nb_rows * W <= 200 * 1920:
        # a strip of a round of waves or two is as slow as its slowest wave: three buffer sets, each copy on its frame's
        # own stream, the host three frames behind (a 1/8 strip of the Cornell frame: 0.038 ms per delivered frame
        # against 0.045 with the whole frame's settings; a quarter of the frame and more is served as well or better by
        # the whole frame's: profiles/r4/readback_routes.txt)
        depth, engine_sets, copy_inline = 4, 3, True
    if os.environ.get("SOLR_BENCH_ENGINE_SETS"):          # experiments (profiles/r4/readback_routes.txt)
        engine_sets = max(1, min(4, int(os.environ["SOLR_BENCH_ENGINE_SETS"])))
    if os.environ.get("SOLR_BENCH_LAG"):
        depth = int(os.environ["SOLR_BENCH_LAG"]) + 1
    if distributed and os.environ.get("SOLR_BENCH_STRIP_SETTINGS") == "1":   # (experiments: a one-rank job with a strip's settings)
        depth, engine_sets, copy_inline = 4, 3, True
    lag = depth - 1
    tickets = deque()
    delivered = [0]
    hip.solr_hip_image_wait.restype = C.c_void_p

    last_image = [None]       # address of the host image the newest delivered frame is in (rank 0)

    def take(ticket):
        if ticket >= 0:
            image = hip.solr_hip_image_wait(ticket)
            if not image:
                k.check(-1, "solr_hip_image_wait")
            last_image[0] = image
            delivered[0] += 1

    def step():
        if pipe is not None:                    # --torch-gather: the older loop, the frame stays on the device
            pipe.frame(render)
            return
        frame()
        tickets.append(hip.solr_hip_d2h_gathered_async() if (native and mode["delivery"] == "gathered")
                       else hip.solr_hip_d2h_image_async())
        if tickets[-1] == -1:
            k.check(-1, "read-back of the frame")
        while len(tickets) > lag:
            take(tickets.popleft())              # (-2 on the ranks that are not the root: nothing to deliver)

    def drain():
        while tickets:
            take(tickets.popleft())
        sync()

    if os.environ.get("SOLR_BENCH_COPY_INLINE"):
        copy_inline = os.environ["SOLR_BENCH_COPY_INLINE"] == "1"

    def use_delivery_pipeline():
        if pipe is None:
            hip.solr_hip_set_frames_in_flight(engine_sets)
            hip.solr_hip_set_copy_route(1 if copy_inline else 0)

    def timed(steps, warmup, regions):
        """W untimed steps, then `regions` regions back to back, each EXACTLY `steps` steps between barrier +
        synchronisation on both sides (a region ends when its last image is on the host).  No event, no instrument
        of any kind inside a region.  Returns every region's wall time."""
        for _ in range(warmup):
            step()
        drain()
        out, own = [], []
        issued = 0.0
        for _ in range(regions):
            barrier()
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            t_issued = time.perf_counter()
            drain()
            own.append(time.perf_counter() - t0)       # this rank's K frames are delivered: its clock stops here
            barrier()
            out.append(time.perf_counter() - t0)       # (with the control plane's barrier: gloo over TCP, not the path's)
            issued += t_issued - t0
        k.check(0, "timed frames")
        # the image the LAST timed step delivered, as the host received it (outside the regions; main() compares it
        # with the oracle's frame: the timed frames are the right frames).  The ring's slot stays untouched until the
        # next step is issued
        snapshot = None
        if rank == 0 and last_image[0] and not cfg4:
            import numpy as np
            snapshot = np.ctypeslib.as_array((C.c_ubyte * (W * H * 3)).from_address(last_image[0])).reshape(H, W, 3).copy()
        # A region's time is the MAX over ranks of `own` (reduce_over_ranks): the moment the slowest rank's last frame
        # was delivered, counted from the start barrier.  The end barrier brackets the region but is not charged to it -
        # a gloo barrier of eight ranks is of the order of a 20-step region of an eight-GPU frame
        return {"regions": own, "with_end_barrier": out, "issued": issued / max(regions, 1), "strip": current_strip(),
                "last_image": snapshot}

    def current_strip():
        a, b = C.c_int(), C.c_int()
        hip.solr_hip_get_strip(C.byref(a), C.byref(b))
        return [a.value, b.value]

    def region_count(steps):
        """at least 25 regions, and enough of them to keep the GPU busy for a fifth of a second: a 20-step region of
        this workload is 6 ms, and neither a median of three nor a utilisation sampler sees that"""
        if os.environ.get("SOLR_BENCH_REGIONS"):
            return max(1, int(os.environ["SOLR_BENCH_REGIONS"]))
        probe = timed(steps, 0, 1)["regions"][0]
        n = max(25, int(0.2 / max(probe, 1e-6)) + 1)
        if distributed:
            t = torch.tensor([float(n)], dtype=torch.float64, device="cpu" if native else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)      # every rank runs the same number of regions
            n = int(t[0])
        return min(n, 400)

    def kernel_segment():
        """the renderer's own duration: HIP events around each of 16 launches issued one at a time, outside every timed
        region (an event pair around an overlapped launch spans parts of two frames)"""
        if pipe is not None:
            return 0.0, None
        hip.solr_hip_set_frames_in_flight(1)
        for _ in range(PREROLL_FRAMES):     # the tile-cost feedback settles on the one buffer set (the order is made anew when the number of frames in flight changes)
            frame()
        sync()
        hip.solr_hip_kernel_time(None, 1)
        hip.solr_hip_enable_timing(1)
        for _ in range(16 if not cfg4 else 74):
            frame()
            sync()
        hip.solr_hip_enable_timing(0)
        samples = (C.c_float * 128)()
        got = hip.solr_hip_timing_samples(samples, None, 128)
        launches = C.c_int(0)
        avg = hip.solr_hip_kernel_time(C.byref(launches), 1) / max(launches.value, 1)
        use_delivery_pipeline()
        return avg, spread(list(samples[:got]))

    # setup, untimed: let the clocks and the tile-cost feedback of the engine settle before the W
    # warmup steps (a frame is 0.4 ms; W = 3 alone is 1.2 ms of GPU work, shorter than the power ramp)
    use_delivery_pipeline()
    for _ in range(PREROLL_FRAMES):
        step()
    drain()
    arm("choosing the number of regions")
    regions = region_count(args.steps)
    second = None
    if balanced:
        # equal strips share out rows, not work (profiles/r2/strip_balance_*.txt): the reference's equal split is
        # timed first, as the second figure; then the strips are re-cut by the cost the frames recorded, the
        # tile-cost feedback settles on the new strips, and the headline segment runs on them
        if world > 1:
            arm("equal strips")
            second = timed(args.steps, args.warmup, max(regions // 3, 5))
        if hip.solr_hip_balance_strips() != 0:
            k.check(-1, "solr_hip_balance_strips")
        for _ in range(PREROLL_FRAMES // 2):
            step()
        drain()
    # ---- N > 1: which communicator mode, which delivery route? (bench_dist.mode_sweep: every combination timed in short
    # segments of this same loop, its frame checked against the one-GPU frame; the headline runs on the fastest that
    # passed AND makes the RCCL call behind every frame)
    mode_sweep = None
    combos = [(c, r, True) for c in sweep_comms for r in sweep_routes]
    if "strips" in sweep_routes and "SOLR_BENCH_COLLECTIVE" not in os.environ:
        # ... and the strips route with NO collective in the data path: the reference assembles the frame in the host
        # bitmap and nowhere else (d2h_bitmap, CudaRayTracer.cu:1647-1672); the shared host image is that.  Timed and
        # reported, never the headline.  The communicator stays up: strips are balanced through it, ambient-occlusion
        # frames trade their boundary rows through it, and the check after the timed regions gathers once.
        combos.append((sweep_comms[0], "strips", False))
    elif not mode["collective"]:
        combos = [(sweep_comms[0], "strips", False)]
    want_sweep = os.environ.get("SOLR_BENCH_SWEEP", "1" if world > 1 else "0") == "1"
    if native and not cfg4 and len(combos) > 1 and want_sweep:
        from types import SimpleNamespace
        loop = SimpleNamespace(step=step, drain=drain, sync=sync, barrier=barrier, timed=timed, tickets=tickets)
        mode_sweep = bench_dist.mode_sweep(R, loop, combos, steps=args.steps, warmup=args.warmup, regions=regions, alone=alone,
                                           balanced=balanced, strip=(first_row, nb_rows), arm=arm, engine_failure=solr.SolrError)
        args.delivery = mode["delivery"]
    if native:
        # `value` at N > 1 is the RCCL-gather route, always: the gather behind every frame, on a communicator of
        # exactly --gpus ranks (a job whose communicator has another size, or none, is void: exit code 5)
        if not (os.environ.get("SOLR_BENCH_COLLECTIVE") == "0"):
            mode["collective"] = True
        rccl_ranks, comm_count = R.require_communicator()
    arm("the timed regions")
    main_run = timed(args.steps, args.warmup, regions)
    if native and mode["collective"] and world > 1 and R.agree(rank == 0 and delivered[0] == 0):
        raise SystemExit("bench.py rank %d: the timed regions delivered no frame to rank 0's host" % rank)
    short_ray_lists = int(hip.solr_hip_short_ray_lists())   # what the engine chose for the delivered frames
    t_issued_ms = main_run["issued"] / args.steps * 1e3
    strips = "balanced by cost" if balanced else "equal rows"
    arm("kernel time")
    kernel_avg_ms, kernel_spread = kernel_segment()
    main_run["kernel_avg_ms"] = kernel_avg_ms
    if second:
        second["kernel_avg_ms"] = kernel_avg_ms
    kernel_basis = ("HIP events around each of %d launches issued one at a time after the timed regions (none inside them)"
                    % (74 if cfg4 else 16))

    arm("frame protocols")
    # ---- the same frame under the reference's own frame protocol (N = 1; untimed extras, reported next to
    # `value`): one frame at a time - render, wait - and cudaRender + d2h_bitmap, the wall time SURVEY.md
    # section 8(d) defines the metric over (kernel + read-back of the RGB image and the primitive ids)
    rates = None
    if not distributed:
        import numpy as np
        hip.solr_hip_set_frames_in_flight(1)
        n_extra = max(8, min(args.steps, 64))
        if cfg4:
            n_extra = 74  # one whole cycle of passes 0...73 (they differ in cost), wherever the timed region stopped
        for _ in range(4):
            frame()
        sync()
        ta = time.perf_counter()
        for _ in range(n_extra):
            frame()
            sync()
        one_at_a_time = (time.perf_counter() - ta) / n_extra
        host_rgb = np.zeros((H, W, 3), np.uint8)
        host_ids = np.zeros((H, W, 4), np.int32)
        hip.solr_hip_d2h(C.byref(si), C.c_void_p(host_rgb.ctypes.data), C.c_void_p(host_ids.ctypes.data))
        ta = time.perf_counter()
        for _ in range(n_extra):
            frame()
            hip.solr_hip_d2h(C.byref(si), C.c_void_p(host_rgb.ctypes.data), C.c_void_p(host_ids.ctypes.data))
        with_d2h = (time.perf_counter() - ta) / n_extra
        ta = time.perf_counter()
        for _ in range(n_extra):
            frame()
            hip.solr_hip_d2h(C.byref(si), C.c_void_p(host_rgb.ctypes.data), None)
        with_image_d2h = (time.perf_counter() - ta) / n_extra
        # ... and with the image leaving in bands while the kernel renders (include/solr_hip.h solr_hip_stream_next_image:
        # what HipKernel's render_begin / render_end and SolR_RunKernel do one frame at a time); frames that cannot be
        # streamed (a neighbourhood post-process, a tile the launch splits) are read back behind the kernel as above
        in_bands = in_bands_ids = None
        if hip.solr_hip_stream_next_image(0) == 1:
            def streamed_frame():
                hip.solr_hip_stream_next_image(1)
                frame()
                if hip.solr_hip_d2h_streamed_image(C.c_void_p(host_rgb.ctypes.data)) != 1:
                    hip.solr_hip_d2h(C.byref(si), C.c_void_p(host_rgb.ctypes.data), None)
            for _ in range(20):                  # (the launch order of a streamed frame settles)
                streamed_frame()
            left_before = hip.solr_hip_stream_next_image(-2)
            ta = time.perf_counter()
            for _ in range(n_extra):
                streamed_frame()
            in_bands = ((time.perf_counter() - ta) / n_extra, hip.solr_hip_stream_next_image(-2) - left_before)
            # ... and the reference's whole d2h_bitmap - image AND primitive ids, 39 MB of a 1080p frame - in bands
            def streamed_frame_with_ids():
                hip.solr_hip_stream_next_image(2)
                frame()
                if hip.solr_hip_d2h_streamed(C.c_void_p(host_rgb.ctypes.data), C.c_void_p(host_ids.ctypes.data)) != 1:
                    hip.solr_hip_d2h(C.byref(si), C.c_void_p(host_rgb.ctypes.data), C.c_void_p(host_ids.ctypes.data))
            for _ in range(4):
                streamed_frame_with_ids()
            left_before = hip.solr_hip_stream_next_image(-2)
            ta = time.perf_counter()
            for _ in range(n_extra):
                streamed_frame_with_ids()
            in_bands_ids = ((time.perf_counter() - ta) / n_extra, hip.solr_hip_stream_next_image(-2) - left_before)
            for _ in range(20):                  # ... and the order by cost alone comes back for what follows
                frame()
            sync()
        # the frames left in HBM, pipelined over the engine's buffer sets (earlier rounds' headline: nothing delivered)
        device_resident = None
        if not cfg4:
            hip.solr_hip_set_frames_in_flight(args.frames_in_flight)
            for _ in range(8):
                frame()
            sync()
            samples = []
            for _ in range(9):
                sync()
                ta = time.perf_counter()
                for _ in range(args.steps):
                    frame()
                sync()
                samples.append((time.perf_counter() - ta) / args.steps)
            device_resident = sorted(samples)[len(samples) // 2]
        k.check(0, "frame protocol rates")
        use_delivery_pipeline()
        rates = {"one_frame_at_a_time": (one_at_a_time, "render, wait for it, render the next"),
                 "cudaRender_plus_d2h_bitmap": (with_d2h, "render + read-back of the RGB image and the primitive ids "
                                                "(the reference's render_begin / render_end, SURVEY.md 8d)"),
                 "cudaRender_plus_image": (with_image_d2h, "render + read-back of the RGB image alone (the ids stay on "
                                           "the device until picking asks: HipKernel::render_end)")}
        if in_bands:
            rates["cudaRender_plus_image_in_bands"] = (
                in_bands[0], "render + the RGB image leaving in bands of tile rows while the kernel renders the rows below "
                "(%d of %d frames did; HipKernel's render_begin / render_end and SolR_RunKernel, one frame at a time)" % (in_bands[1], n_extra))
        if in_bands_ids:
            rates["cudaRender_plus_d2h_bitmap_in_bands"] = (
                in_bands_ids[0], "render + RGB image and primitive ids leaving in bands while the kernel renders (%d of %d frames did; "
                "solr_hip_stream_next_image(2) + solr_hip_d2h_streamed)" % (in_bands_ids[1], n_extra))
        if device_resident:
            rates["pipelined_device_resident"] = (
                device_resident, "%d frames in flight, the image left in HBM, nothing delivered (what earlier rounds "
                "reported as `value`; median of 9 regions of %d steps)" % (args.frames_in_flight, args.steps))

    # ---- the walk's own ceiling (SURVEY.md 8d's second yardstick; N = 1, untimed extra): the walks of one recorded
    # frame replayed with nothing but the node loop (include/solr_hip.h solr_hip_walk_bound)
    walk_bound = None
    if not distributed and not cfg4 and not args.no_walk_bound:
        arm("the walk's own ceiling")
        hip.solr_hip_set_frames_in_flight(1)
        for _ in range(24):                  # the launch order the timed frames had settles again on one buffer set
            frame()
        sync()
        ms3 = (C.c_double * 3)()
        stats = (C.c_ulonglong * 4)()
        if hip.solr_hip_walk_bound(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles), 20,
                                   ms3, stats) == 0:
            walk_bound = {"replay_ms": float(ms3[1]), "replay_ms_min": float(ms3[2]), "walks_recorded_per_wave": int(stats[0]),
                          "walks_not_replayed": int(stats[1]), "leaf_entries_per_lane": int(stats[2]),
                          "workgroups": int(stats[3])}
        else:
            buf = C.create_string_buffer(512)
            hip.solr_hip_last_error(buf, 512)
            walk_bound = {"error": buf.value.decode(errors="replace")}
            hip.solr_hip_clear_error()
        use_delivery_pipeline()

    # ---- N > 1 extras (untimed): the gather alone, and the assembled frame against the frame one GPU renders
    gather_only_ms = None
    check = None
    headline_collective = mode["collective"]      # (what the timed regions ran with; the extras below gather)
    mode["collective"] = True
    if native:
        arm("the gather alone and the one-GPU check")
        import numpy as np
        for _ in range(8):
            hip.solr_hip_gather_strips(0)
        sync()
        barrier()
        ta = time.perf_counter()
        for _ in range(64):
            hip.solr_hip_gather_strips(0)
        sync()
        barrier()
        gather_only_ms = (time.perf_counter() - ta) / 64 * 1e3
        k.check(0, "gather alone")
        if not args.no_check:
            # one more frame (cfg4: one more cycle of passes 0...73) on the strips of the timed region, gathered;
            # then everybody leaves the communicator and rank 0 renders the same frame alone, whole
            keep = {}
            passes = range(74) if cfg4 else range(1)
            pass_counter[0] = 0
            for it in passes:
                frame()
                if rank == 0 and it in (0, 11, 73):
                    got = np.zeros((H, W, 3), np.uint8)
                    if hip.solr_hip_d2h_gathered(C.c_void_p(got.ctypes.data)) != 0:
                        k.check(-1, "solr_hip_d2h_gathered")
                    keep[it] = got
            sync()
            barrier()
            hip.solr_hip_comm_finalize()
            if rank == 0:
                hip.solr_hip_set_strip(0, -1)
                hip.solr_hip_set_frames_in_flight(1)
                pass_counter[0] = 0
                same = True
                for it in passes:
                    render()
                    if it in keep:
                        alone = np.zeros((H, W, 3), np.uint8)
                        hip.solr_hip_d2h(C.byref(si), C.c_void_p(alone.ctypes.data), None)
                        same = same and bool(np.array_equal(alone, keep[it]))
                k.check(0, "the frame on one GPU")
                check = same

    rays_total = rays_local
    per_rank = None
    second_out = None
    region_times = list(main_run["regions"])
    second_times = list(second["regions"]) if second else []
    if distributed:
        # every region's time is the slowest rank's (MAX over ranks, region by region); rays are summed
        region_times, second_times, kernel_avg_ms, rays_total = bench_dist.reduce_over_ranks(
            dist, torch, region_times, second_times, kernel_avg_ms, rays_local, "cpu" if native else "cuda")
        mine = {"rank": rank, "rows": main_run["strip"], "ms_per_step_until_own_frames_delivered":
                round(median(main_run["regions"]) / args.steps * 1e3, 4), "host_issue_ms_per_step": round(t_issued_ms, 4),
                "kernel_ms": round(main_run["kernel_avg_ms"], 5)}
        if second:
            mine["equal_strips"] = {"rows": second["strip"], "ms_per_step_until_own_frames_delivered":
                                    round(median(second["regions"]) / args.steps * 1e3, 4)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        if second:
            second_out = median(second_times)
        # every rank empties its C stdout buffer (RCCL's version banner sits there until exit) before rank 0
        # may print: the JSON line is then the last thing the job writes to stdout
        C.CDLL(None).fflush(None)
        sys.stdout.flush()
        dist.barrier()

    disarm()
    if native:
        hip.solr_hip_comm_finalize()
    if rank != 0:
        dist.destroy_process_group()
        C.CDLL(None).fflush(None)
        return

    # the headline: the MEDIAN region (each exactly --steps frames between barrier + synchronisation), with the
    # fastest and the slowest region as its error bar - which brackets it by construction
    elapsed = median(region_times)
    step_spread = {"min": round(min(region_times) / args.steps * 1e3, 5), "median": round(elapsed / args.steps * 1e3, 5),
                   "max": round(max(region_times) / args.steps * 1e3, 5), "regions": len(region_times),
                   "steps_per_region": args.steps,
                   "timed_seconds_in_all": round(sum(region_times), 4)}
    mrays = rays_total * args.steps / elapsed / 1e6
    scene_bytes = (48 * len(flat.boxes) + 128 * len(flat.primitives) + 48 * len(flat.lights) +
                   176 * len(set(int(m) for m in flat.primitives["materialId"])))
    # per launch, rank 0's strip: pass 0 writes only; cfg4 cycles through 1 first pass and 73 later ones
    per_pixel = FIRST_PASS_BYTES_PER_PIXEL if not cfg4 else (FIRST_PASS_BYTES_PER_PIXEL + 73 * LATER_PASS_BYTES_PER_PIXEL) / 74.0
    rows_rank0 = main_run["strip"][1] if distributed else nb_rows
    algo_bytes = int(rows_rank0 * W * per_pixel) + scene_bytes
    achieved = algo_bytes / (kernel_avg_ms * 1e-3) / 1e9 if kernel_avg_ms > 0 else 0.0
    # the same launch priced with SURVEY.md 8(d)'s fused figure (67 B per pixel: it counts the ids read of the
    # early-out test, which a first pass does not make)
    survey_bytes = int(rows_rank0 * W * (SURVEY_FUSED_BYTES_PER_PIXEL if not cfg4 else per_pixel)) + scene_bytes
    achieved_survey = survey_bytes / (kernel_avg_ms * 1e-3) / 1e9 if kernel_avg_ms > 0 else 0.0
    traffic, traffic_source, traffic_commit = measured_traffic(args, world)
    valu = measured_counters(args, world)
    label = {"cornell": "Cornell", "height_field": "triangle mesh", "molecule": "molecule"}.get(args.scene, args.scene)
    metric = "Mrays/s @%dx%d, %d-bounce %s" % (W, H, si.nbRayIterations, label)
    if cfg4:
        metric = "Mrays/s @3840x2160, Cornell, passes 0-73 (64 accumulated samples, depth of field + ambient occlusion)"
    out = {
        "metric": metric,
        "value": round(mrays, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "regions": len(region_times),
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        # which of config.rates_mrays_per_s `value` is: the frame DELIVERED (its image on the host every frame, the host
        # `frames in flight - 1` frames behind the renderer; N > 1: behind the RCCL gather of the strips on every frame) -
        # not the reference's one-frame-at-a-time protocol (rates: one_frame_at_a_time, cudaRender_plus_image,
        # cudaRender_plus_d2h_bitmap) and not frames left in HBM (pipelined_device_resident)
        "protocol": "delivered_pipelined" if not distributed else (
            "delivered_pipelined_behind_the_rccl_gather" if (native and headline_collective) else
            ("delivered_pipelined_no_collective (SOLR_BENCH_COLLECTIVE=0)" if native else "torch_gather_device_resident")),
        "dtype": "f32",
        "data": "synthetic" if args.scene not in ("irt_model", "obj_model", "swc_morphology", "pdb_molecule") else "the reference's sample scene file",
        "config": {"workload": "%s%s %dx%d, %d bounces + shadow rays, %d boxes / %d primitives, %s" %
                   ((args.config + ": ") if args.config else "", args.scene, W, H, si.nbRayIterations, len(flat.boxes),
                    len(flat.primitives), ("%d row strips, %s" % (world, strips)) if distributed else "one GPU, whole frame"),
                   "rays_per_frame": rays_total, "closest_hit_walks_rank0": int(counts[0]),
                   "shadow_walks_rank0": int(counts[1]), "lane_nodes": int(counts[2]), "lane_prim_tests": int(counts[3]),
                   "wave_nodes": int(counts[4]), "wave_prim_tests": int(counts[5]), "wave_walks": int(counts[6]) + int(counts[7]), "mpixels_per_s": round(W * H * args.steps / elapsed / 1e6, 2),
                   "host_issue_ms_per_step_rank0": round(t_issued_ms, 4),
                   "cost_ordered_launch_rank0": bool(hip.solr_hip_tile_scheduling_active()),
                   "frames_in_flight": args.frames_in_flight,
                   # untimed set-up frames before the W warm-up steps (clocks, tile-cost feedback, order-free lists)
                   "setup_frames_before_warmup": PREROLL_FRAMES,
                   # the error bar of `ms_per_step`: the timed region - exactly `steps` frames delivered, between barrier +
                   # synchronisation - is run `regions` times back to back; ms_per_step is the median region, min / max
                   # the fastest and the slowest one (ms per step each)
                   "step_ms_spread": step_spread,
                   "delivery": ("frames left on the device (--torch-gather)" if pipe is not None else
                                "every frame's image lands on the host: page-locked ring, copy stream behind the %s, "
                                "the host %d frame(s) behind; %d buffer set(s) in the engine; primitive ids on demand "
                                "(d2h_bitmap when picking asks)" %
                                (("gather on rank 0 (the assembled frame over rank 0's PCIe link)" if args.delivery == "gathered"
                                  else "kernel on every rank: each strip over its rank's own PCIe link into one image the "
                                       "ranks' processes share (solr_hip_image_share), rank 0 waits for all of them" +
                                       ("; the RCCL gather assembles the same frame in rank 0's HBM" if headline_collective
                                        else "; nothing else moves per frame")) if native else "kernel",
                                 lag, engine_sets)) + ("; copies on the frames' own streams" if copy_inline else ""),
                   "frames_delivered": delivered[0], "delivery_fallback": delivery_fallback,
                   # nodes per order-free list when long rays' walks use them, else 0
                   "order_free_nodes": int(hip.solr_hip_order_free_nodes()),
                   # 1: bounce rays of the long-list triangle kernels took them too during the timed regions (the engine's
                   # choice per frame, include/solr_hip.h solr_hip_set_short_ray_lists; nothing to choose for other scenes)
                   "short_ray_lists": short_ray_lists,
                   "gather": ("none (one GPU)" if not distributed else
                              ("RCCL from the engine's C ABI (solr_hip_gather_strips), on the stream that rendered the frame"
                               if headline_collective else
                               "none in the data path: the strips meet in the host image the ranks' processes share (the "
                               "reference's d2h_bitmap assembles the frame there and nowhere else); RCCL balanced the strips "
                               "and assembled one frame in rank 0's HBM for gathered_equals_single_gpu, outside the timed regions")
                              if native else "torch.distributed gather (RCCL)"),
                   "strips": strips,
                   "parallelism": "tile%d" % world},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_taken_at_commit": traffic_commit,
                     "traffic_note": "read from a committed rocprofv3 --pmc profile of this same command (separate FETCH_SIZE / "
                                     "WRITE_SIZE passes, corrected by the factors measured on known byte counts in the "
                                     "renderer's own access patterns: profiles/r6/fetch_calibration.txt), not measured by "
                                     "this run (a process cannot profile itself); null when the workload differs from "
                                     "the profiled one",
                     "kernel": "k_standardRenderer", "kernel_ms": round(kernel_avg_ms, 5), "kernel_ms_basis": kernel_basis,
                     "kernel_ms_spread": kernel_spread,
                     "algorithmic_bytes": algo_bytes, "algorithmic_bytes_per_pixel": round(per_pixel, 2),
                     "frac_with_survey_8d_bytes": round(achieved_survey / HBM_PEAK_GBPS, 6),
                     "survey_8d_bytes_per_pixel": SURVEY_FUSED_BYTES_PER_PIXEL if not cfg4 else round(per_pixel, 2)},
    }
    if distributed:
        slowest = max(per_rank, key=lambda r: r["ms_per_step_until_own_frames_delivered"])
        out["config"].update({
            "rccl_ranks": rccl_ranks, "rccl_communicators": comm_count,
            "rccl_communicator_mode": ("one per frame in flight" if comm_count and comm_count > 1 else "one for everything") +
                                      (" (the fastest combination of config.mode_sweep that gathers with RCCL behind every frame)" if mode_sweep else
                                       " (SOLR_HIP_COMM_PER_FLIGHT=0|1 / --delivery chose it: no sweep)"),
            "per_rank": per_rank, "slowest_rank": slowest["rank"],
            # the end barrier of a region belongs to the control plane (gloo over TCP) and is not charged to the region:
            # a region's time is the MAX over ranks of each rank's own time from the start barrier to its last delivery
            "control_plane_barrier_ms_rank0": round(1e3 * median([b - a for a, b in zip(main_run["regions"], main_run["with_end_barrier"])]), 4),
            # (for comparison: the median region on rank 0's clock WITH the end barrier inside it - after a barrier every
            # rank's clock reads about the same)
            "ms_per_step_with_the_end_barrier_rank0": round(1e3 * median(main_run["with_end_barrier"]) / args.steps, 4),
            "gather_only_ms": round(gather_only_ms, 4) if gather_only_ms is not None else None,
            "gather_only_note": "64 gathers of the last strips back to back on one stream, no rendering in between",
            # the assembled frame of the strips of the timed region against the same frame rendered whole by rank 0
            # alone (cfg4: after passes 0, 11 and 73 of a cycle, through the depth-halo exchange), RGB8 bit for bit
            "gathered_equals_single_gpu": check,
            "rates_mrays_per_s": {("balanced_strips" if balanced else "equal_strips") +
                                  ("_native_gather" if native else "_torch_gather"): round(mrays, 1)}})
        if mode_sweep:
            # every communicator mode x delivery route, timed in short segments of the same loop before the headline
            # (which ran on the fastest whose frame equals the one-GPU frame); a combination that failed says why
            out["config"]["mode_sweep"] = mode_sweep
            for name, entry in mode_sweep.items():
                if isinstance(entry, dict) and "ms_per_step" in entry:
                    entry["mrays_per_s"] = round(rays_total / (entry["ms_per_step"] * 1e-3) / 1e6, 1)
                    out["config"]["rates_mrays_per_s"]["sweep_" + name] = entry["mrays_per_s"]
        if second_out:
            out["config"]["rates_mrays_per_s"]["equal_strips_native_gather"] = round(rays_total * args.steps / second_out / 1e6, 1)
            out["config"]["rates_note"] = {"equal_strips_native_gather": "the reference's equal split, same loop, timed "
                                           "before the strips were re-cut: %.4f ms per step" % (second_out / args.steps * 1e3)}
        if share:
            out["config"]["rehearsal"] = ("SOLR_BENCH_SHARE_GPU=1: every rank on GPU 0 - a rehearsal of the N > 1 code "
                                          "path, not a measurement of N GPUs")
    if rates:
        out["config"]["rates_mrays_per_s"] = {"delivered_pipelined": round(mrays, 1)}
        out["config"]["rates_note"] = {"delivered_pipelined": "`value`: every frame's image on the host, the host %d "
                                       "frame(s) behind the renderer" % lag}
        for name, (seconds, what) in rates.items():
            out["config"]["rates_mrays_per_s"][name] = round(rays_total / seconds / 1e6, 1)
            out["config"]["rates_note"][name] = "%s: %.4f ms per frame" % (what, seconds * 1e3)
    if walk_bound and "replay_ms" in walk_bound and walk_bound["replay_ms"] > 0:
        # the frame's walks with nothing but the node loop (the same waves, rays and lists; shadow walks node for node,
        # closest-hit walks with their final cut-off from the first node): what the walk STRUCTURE costs, and the rate
        # a renderer whose leaves and shading were free would reach on this frame
        bound = rays_total / (walk_bound["replay_ms"] * 1e-3) / 1e6
        out["roofline"]["walk_bound_mrays"] = round(bound, 1)
        out["roofline"]["walk_bound"] = {
            "node_loop_only_ms": round(walk_bound["replay_ms"], 5), "node_loop_only_ms_min": round(walk_bound["replay_ms_min"], 5),
            "frac_of_bound_achieved": round(mrays / bound, 4),
            "frac_of_bound_achieved_by_the_kernel_alone": round(walk_bound["replay_ms"] / kernel_avg_ms, 4) if kernel_avg_ms > 0 else None,
            "walks_recorded_per_wave": walk_bound["walks_recorded_per_wave"],
            "walks_not_replayed": walk_bound["walks_not_replayed"],
            "leaf_entries_per_lane": walk_bound["leaf_entries_per_lane"], "workgroups": walk_bound["workgroups"],
            "note": "solr_hip_walk_bound: one frame's walks recorded (list, ray, cut-off, when a shadow lane was done), "
                    "replayed 20 times with the node loop alone at the renderer's occupancy, HIP events around each launch"}
    elif walk_bound:
        out["roofline"]["walk_bound_mrays"] = None
        out["roofline"]["walk_bound"] = walk_bound
    if valu and kernel_avg_ms > 0:
        # instruction-issue view of the same launch (informative; the contract's roofline is the HBM one): vector
        # instructions against 4 SIMD-32 per CU x one wave64 instruction per 2 cycles, scalar instructions against
        # one scalar unit per CU x one per cycle (MI355X_MICROARCH.md), and against what tools/valu_issue_bench.hip
        # measured on this chip (profiles/r2/valu_issue_bench.txt: 0.885e12 v_fma_f32, 0.575e12 s_add_i32 per second)
        seconds = kernel_avg_ms * 1e-3
        out["roofline"]["issue"] = {
            "source": valu["source"], "taken_at_commit": valu.get("commit"),
            "vector_insts_per_launch": valu["SQ_INSTS_VALU"], "vector_peak_per_s": VALU_PEAK_WAVE_INSTS_PER_S,
            "vector_frac": round(valu["SQ_INSTS_VALU"] / seconds / VALU_PEAK_WAVE_INSTS_PER_S, 4),
            "vector_frac_of_measured_peak": round(valu["SQ_INSTS_VALU"] / seconds / 0.885e12, 4),
            "scalar_insts_per_launch": valu["SQ_INSTS_SALU"] + valu.get("SQ_INSTS_SMEM", 0),
            "scalar_peak_per_s": SALU_PEAK_WAVE_INSTS_PER_S,
            "scalar_frac": round((valu["SQ_INSTS_SALU"] + valu.get("SQ_INSTS_SMEM", 0)) / seconds / SALU_PEAK_WAVE_INSTS_PER_S, 4),
            "note": "counters from a committed profile of this command; durations from this run"}

    # ---- proof that the timed frames are the right frames (untimed): the image the last timed step delivered against
    # the frame the CPU oracle renders of the same scene, and the engine's ray census against the oracle's own count
    # ("counted by an instrumented build of our CPU restatement")
    rc = 0
    if not args.no_check and pipe is None:
        image, first_pass_rays = main_run.get("last_image"), rays_total
        if cfg4 and not distributed:
            # a step is a pass and the last delivered image is pass 73's: one pass is checked - pass 0, rendered and
            # read back once more after the regions - against the oracle's pass 0 with the same post-processing
            import numpy as np
            hip.solr_hip_set_frames_in_flight(1)
            pass_counter[0] = 0
            render()
            image = np.zeros((H, W, 3), np.uint8)
            hip.solr_hip_d2h(C.byref(si), C.c_void_p(image.ctypes.data), None)
            k.check(0, "the check frame")
            si.pathTracingIteration = 0
            first_pass_rays = rays_first_pass_local
        if image is not None and os.environ.get("SOLR_BENCH_TAMPER") == "1":
            image = image.copy()             # (tests/test_bench_launcher.py: a wrong frame must fail the job)
            image[H // 2, W // 3:W // 3 + 16] ^= 0x40
        if image is not None:
            proof = prove_frames(flat, si, ppi, eye, direction, angles, image, first_pass_rays)
            out["config"].update(proof)
            if proof["rays_equal_oracle_count"] and not cfg4:
                out["config"]["rays_per_frame"] = proof["oracle_rays_per_frame"]   # the oracle's count IS the figure
            if proof["delivered_frame_equals_oracle"] is False or proof["rays_equal_oracle_count"] is False:
                print("bench.py: the delivered frame or the ray census is NOT the oracle's: %s" % json.dumps(proof),
                      file=sys.stderr, flush=True)
                rc = 4
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(flat, si, ppi, eye, direction, angles, args.cpu_seconds)
    if distributed:
        # RCCL writes its version banner to the C library's stdout buffer when the communicator comes up; on a
        # pipe that buffer is only flushed at exit, after our line.  Tear the group down and flush it first:
        # the JSON line is the last thing on stdout
        dist.destroy_process_group()
    C.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)
    if rc:
        sys.exit(rc)     # the line is printed (what was measured, and that it was not the right frame); the job failed


MAX_MARKED_PIXELS = 8    # tests/test_baseline_sizes.py MAX_EXCEPTIONS: pixels of a frame one RGB8 step off, each behind a mis-rounded powf


def prove_frames(flat, si, ppi, eye, direction, angles, image, engine_rays):
    """The image a timed step delivered against the oracle's frame of the same scene and camera - RGB8 equal on every
    pixel but at most MAX_MARKED_PIXELS, each one step off and each marked by the oracle as having gone through a libm
    result that is not the correctly rounded value (oracle_set_misround_mask; the engine rounds the binary64 result
    once: DESIGN.md section 2.3) - and the engine's ray census against the rays the oracle counted while rendering it.
    Returns the keys that go into `config`; a value is None (with `frame_check_error`) when the check itself could not
    run - that does not fail the job, a frame that differs does."""
    import numpy as np
    try:
        from oracle import loader
        assert not loader.lib().oracle_get_rounded_transcendentals()
        misround = np.zeros((si.size_y, si.size_x), np.uint8)
        t0 = time.perf_counter()
        _, _, orgb, counts, status = loader.render(flat, si, ppi, eye, direction, angles, misround=misround)
        seconds = time.perf_counter() - t0
    except Exception as e:       # noqa: BLE001 - the oracle library missing or unloadable: reported, not fatal
        return {"delivered_frame_equals_oracle": None, "rays_equal_oracle_count": None,
                "frame_check_error": "%s: %s" % (type(e).__name__, e)}
    oracle_rays = int(counts[0]) + int(counts[1])
    image = np.asarray(image).reshape(orgb.shape)
    off = np.abs(image.astype(np.int16) - orgb.astype(np.int16)).max(axis=-1)
    differing = off > 0
    unmarked = differing & (misround == 0)
    same = bool(status == 0 and not unmarked.any() and int(differing.sum()) <= MAX_MARKED_PIXELS and int(off.max()) <= 1)
    return {"delivered_frame_equals_oracle": same, "rays_equal_oracle_count": bool(status == 0 and oracle_rays == int(engine_rays)),
            "oracle_rays_per_frame": oracle_rays, "engine_census_rays_per_frame": int(engine_rays),
            "frame_check": {"pixels": int(off.size), "pixels_differing": int(differing.sum()),
                            "of_which_not_behind_a_misrounded_libm_result": int(unmarked.sum()),
                            "largest_rgb8_difference": int(off.max()), "pixels_the_oracle_marked": int((misround != 0).sum()),
                            "allowed_marked_pixels": MAX_MARKED_PIXELS, "oracle_seconds": round(seconds, 2),
                            "what": "the image the last timed step delivered to the host against the CPU oracle's frame "
                                    "(as pinned: libm's binary32 powf), whole frame"}}


def _profiled_workload(args, world):
    """the key of this workload in the committed profiles (a scene name, or cfg4), None when it was not profiled"""
    if world != 1 or args.graphics_level != 4:
        return None
    if args.config == "cfg4":
        return "cfg4"
    if (args.width, args.height) != (1920, 1080) or (args.scene == "cornell" and args.iterations != 3):
        return None
    return args.scene


def measured_traffic(args, world):
    """HBM-side bytes per launch of the renderer from the committed rocprofv3 --pmc passes of this same
    command (profiles/rNN/hbm_traffic.json, written by tools/collect_profiles.py), the file it came from and the
    commit the profile was taken at; None when the workload differs from the profiled one."""
    import glob
    key = _profiled_workload(args, world)
    if not key:
        return None, None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "hbm_traffic.json")), reverse=True):
        try:
            data = json.load(open(path))
        except (OSError, ValueError):
            continue
        entry = data.get(key)
        if entry:
            return entry["bytes_per_launch"], os.path.relpath(path, ROOT), entry.get("commit", data.get("commit"))
    return None, None, None


def measured_counters(args, world):
    """SQ instruction counters per launch of the renderer from the committed PMC passes of this same command
    (profiles/rNN/pmc_<scene>.txt), or None."""
    import glob
    key = _profiled_workload(args, world)
    if not key:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_%s.txt" % key)), reverse=True):
        found = {"source": os.path.relpath(path, ROOT)}
        try:
            for line in open(path):
                if line.startswith("# commit"):
                    found["commit"] = line.split()[-1]
                parts = line.split()
                if len(parts) >= 2 and parts[0] in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"):
                    found[parts[0]] = int(float(parts[1]))
        except (OSError, ValueError):
            continue
        if "SQ_INSTS_VALU" in found and "SQ_INSTS_SALU" in found:
            return found
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
    sample of the same frame.  Thread count: the best SUSTAINED rate (at least 1.5 s each, after a warm-up
    frame: a short burst is not what a CFS-throttled container sustains) of {quota, 2x quota, 4x quota, all
    hardware threads}; then a run of about budget_s seconds with it, and a one-thread figure."""
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

    def frames_for(threads, seconds, rows=H):
        """(rays per second, frames, seconds) of back-to-back frames (of `rows` rows) for at least `seconds`"""
        n, rays, t0 = 0, 0, time.perf_counter()
        while True:
            L.oracle_render(C.byref(scene.c), C.addressof(si), C.addressof(ppi), eye.ctypes.data, direction.ctypes.data,
                            angles.ctypes.data, 0, rows, pp.ctypes.data, ids.ctypes.data, rgb.ctypes.data,
                            C.addressof(counts), threads)
            rays += int(counts[0]) + int(counts[1])
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 4000:
                return rays / dt, n, dt

    hw = len(os.sched_getaffinity(0))
    quota = usable_cpus()
    candidates = sorted(set(min(hw, c) for c in (quota, 2 * quota, 4 * quota, hw)))
    calib = {}
    for c in candidates:
        frames_for(c, 0.0)                       # warm-up frame: thread creation, first touch
        calib[c] = frames_for(c, 1.5)[0]
    threads = max(calib, key=calib.get)
    rate, frames, dt = frames_for(threads, budget_s)
    single = frames_for(1, 2.5)
    return {"value": round(rate / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "threads": threads, "quota_cpus": quota, "hardware_threads": hw,
            "single_thread_value": round(single[0] / 1e6, 3),
            "sample": "%d full %dx%d frames of the same scene in %.1f s, OpenMP over rows with %d threads (affinity "
                      "%d hardware threads, CFS quota %d CPUs; sustained Mrays/s by thread count over >= 1.5 s each: %s); "
                      "one thread: %d frames in %.1f s" %
                      (frames, W, H, dt, threads, hw, quota, {k: round(v / 1e6, 1) for k, v in calib.items()},
                       single[1], single[2])}


if __name__ == "__main__":
    main()
