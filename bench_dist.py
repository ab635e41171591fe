#!/usr/bin/env python3
"""bench_dist.py - the N > 1 side of bench.py: the launcher of a bare `python bench.py --gpus N`, what the ranks agree on
over the control plane (torch.distributed's gloo group: rendezvous, barrier, reductions - never the data path), the
communicator and the delivery route, and the sweep over communicator mode x delivery route that precedes the headline.

The data path is the engine's own (include/solr_hip.h): every rank renders its row strip, `solr_hip_gather_strips` -
one grouped ncclSend per rank, one ncclRecv per peer on the root, enqueued by the library on the stream that rendered the
frame - assembles the frame in rank 0's HBM, and the frame reaches the host either strip by strip over every rank's own
PCIe link (one page-locked image the ranks' processes share) or from rank 0 alone.  BASELINE.json's north_star names "a
single RCCL gather over xGMI to assemble the final image": `value` at N > 1 is always a combination that makes that RCCL
call behind every frame.  Nothing here has run on more than one GPU; tests/test_bench_launcher.py
rehearses it with two ranks on one GPU over tests/loopback_rccl.c.
"""
import ctypes as C
import json
import os
import sys
import time

PREROLL_FRAMES = 48    # (bench.py's: untimed set-up frames; a combination of the sweep settles on half of them)


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def launch(args, script):
    """`python bench.py --gpus N` started bare: this process becomes the launcher - it never touches the GPU - and
    starts N fresh rank processes of this same command (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, exactly what torch.distributed.run would set).  Rank 0's stdout is relayed (its last line is the
    JSON line), the other ranks' goes to stderr.  Any rank that ends with an error ends the job: the others get ten
    seconds, then are killed by pid, and the launcher exits with that rank's code."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    share = os.environ.get("SOLR_BENCH_SHARE_GPU") == "1"
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if share else r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("SOLR_BENCH_TIMEOUT", "1500"))
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            failed = bad[0] if bad else (-1, 124)
            grace = time.time() + 10.0
            while time.time() < grace and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    reader.join(timeout=5.0)
    sys.stdout.write("".join(x.decode(errors="replace") if isinstance(x, bytes) else x for x in lines))
    sys.stdout.flush()
    bad = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
    if failed or bad:
        r, code = failed or bad[0]
        print("bench.py: %s; the job is void" % ("rank %d ended with code %s" % (r, code) if r >= 0 else
                                                   "no result within SOLR_BENCH_TIMEOUT"), file=sys.stderr)
        return code if isinstance(code, int) and 0 < code < 256 else 1
    return 0


def reduce_over_ranks(dist, torch, region_times, second_times, kernel_ms, rays_local, device="cpu"):
    """What the ranks of an N > 1 job agree on after the timed regions: every region's time is the SLOWEST rank's
    (MAX over ranks, region by region - the frame is delivered when the last strip is), the kernel time the slowest
    rank's, the rays the sum.  Returns (region_times, second_times, kernel_ms, rays_total)."""
    t = torch.tensor(list(region_times) + list(second_times) + [kernel_ms], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n, m = len(region_times), len(second_times)
    r = torch.tensor([float(rays_local)], dtype=torch.float64, device=device)
    dist.all_reduce(r, op=dist.ReduceOp.SUM)
    return [float(x) for x in t[:n]], [float(x) for x in t[n:n + m]], float(t[-1]), int(r[0])


def agree(dist, torch, failed_here, device="cpu"):
    """True when ANY rank reports a failure: a decision every rank takes alike (e.g. the delivery route when one of
    them cannot open the shared host image)"""
    flag = torch.tensor([1.0 if failed_here else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return float(flag[0]) > 0


class Ranks:
    """The ranks of one job as every rank sees them: the control plane (dist, torch), the engine library, who I am, and
    what is up right now - `mode["per_flight"]`: one communicator per frame in flight (None: no communicator),
    `mode["delivery"]`: how the frame reaches the host ('strips' / 'gathered'; None: no route), `mode["collective"]`:
    the RCCL gather runs behind every frame."""

    def __init__(self, dist, torch, hip, rank, world, collective=True):
        self.dist, self.torch, self.hip, self.rank, self.world = dist, torch, hip, rank, world
        self.mode = {"per_flight": None, "delivery": None, "collective": collective}

    def agree(self, failed_here):
        return agree(self.dist, self.torch, failed_here)

    def engine_error(self):
        buf = C.create_string_buffer(512)
        self.hip.solr_hip_last_error(buf, 512)
        return buf.value.decode(errors="replace")

    def comm_up(self, per_flight):
        """(every rank) the communicator - rank 0's id to everybody over the control plane, then ncclCommInitRank in the
        library - with one communicator for everything or one per frame in flight (None: as the library / the
        environment says).  False, on every rank alike and with the engine's error cleared, when it did not come up."""
        hip, dist, rank, world, mode = self.hip, self.dist, self.rank, self.world, self.mode
        if per_flight is not None:
            hip.solr_hip_comm_set_per_flight(1 if per_flight else 0)
        uid = C.create_string_buffer(128)
        fine = not (rank == 0 and hip.solr_hip_comm_unique_id(uid) != 0)
        box = [uid.raw]
        dist.broadcast_object_list(box, src=0)
        uid = C.create_string_buffer(box[0], 128)
        fine = fine and hip.solr_hip_comm_init(rank, world, uid) == 0
        if self.agree(not fine):
            if rank == 0:
                print("bench.py: the communicator did not come up (per flight: %s): %s" %
                      (per_flight, self.engine_error() or "on another rank"), file=sys.stderr, flush=True)
            hip.solr_hip_clear_error()
            hip.solr_hip_comm_finalize()
            # nothing is up now: the next configure() must bring a communicator (and a delivery route) up again,
            # whatever it asks for (a stale `False` here let the next combination run without one)
            mode["per_flight"] = None
            mode["delivery"] = None
            return False
        mode["per_flight"] = int(hip.solr_hip_comm_count()) > 1
        return True

    def delivery_up(self, route):
        """(every rank) 'strips': one host image for all ranks - rank 0 creates the segment, the others open it.  A box
        that does not let the processes share page-locked memory (no /dev/shm, a registration the driver refuses) must
        not void the job: all ranks then fall back, together, to rank 0 copying the gathered frame.  Returns the route
        that is up and, when it is not the one asked for, why."""
        hip, dist, rank, world, mode = self.hip, self.dist, self.rank, self.world, self.mode
        hip.solr_hip_image_unshare()
        why = None
        if route == "strips":
            name = ("/solr_bench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getuid())).encode()
            mine = 0
            if os.environ.get("SOLR_BENCH_FAIL_SHARE") == "1" and rank == world - 1:
                name = b"no-leading-slash"          # (tests: the last rank cannot open the segment)
            if rank == 0:
                mine = hip.solr_hip_image_share(name, rank, world)
            dist.barrier()
            if rank != 0:
                mine = hip.solr_hip_image_share(name, rank, world)
            if self.agree(mine != 0):
                why = "solr_hip_image_share failed on some rank (%s): --delivery gathered instead" % \
                    (self.engine_error() or "on another rank")
                hip.solr_hip_clear_error()
                hip.solr_hip_image_unshare()
                route = "gathered"
                mode["collective"] = True      # (that route delivers what the gather assembled: no gather, no frame)
                if rank == 0:
                    print("bench.py: " + why, file=sys.stderr, flush=True)
            dist.barrier()
            # every rank has the segment mapped: its NAME can go (a job that dies later leaves nothing in /dev/shm)
            hip.solr_hip_image_share_sealed()
        mode["delivery"] = route
        return route, why

    def require_communicator(self):
        """(every rank) before the headline: a communicator of exactly `world` ranks is up, or the job is void (exit code
        5): `value` at N > 1 is the RCCL-gather route.  Returns (ranks of the communicator, communicators)."""
        hip = self.hip
        if int(hip.solr_hip_comm_count()) < 1 and not self.comm_up(False):
            raise SystemExit("bench.py rank %d: no communicator for the headline" % self.rank)
        ranks, count = int(hip.solr_hip_comm_ranks()), int(hip.solr_hip_comm_count())
        if self.agree(ranks != self.world):
            if self.rank == 0:
                print("bench.py: the communicator has %d rank(s), --gpus is %d: the job is void" % (ranks, self.world),
                      file=sys.stderr, flush=True)
            sys.exit(5)
        return ranks, count


def label(per_flight, route, collective):
    if not collective:
        return "no_collective_strips_over_every_ranks_link"
    return "%s_%s" % ("communicator_per_flight" if per_flight else "one_communicator",
                      "strips_over_every_ranks_link" if route == "strips" else "gathered_frame_over_rank0s_link")


def mode_sweep(R, loop, combos, *, steps, warmup, regions, alone, balanced, strip, arm, engine_failure):
    """N > 1: which communicator mode, which delivery route?  Nobody has run this on eight GPUs: RCCL orders the
    operations of ONE communicator whatever streams they are enqueued on (one per frame in flight avoids that, at the
    price of four communicators), and the frame can reach the host over every rank's own PCIe link (one shared host
    image) or over rank 0's alone (the gathered frame).  So the job times every combination - short segments of the same
    loop as the headline (`loop`: step, drain, sync, barrier, timed, tickets) - checks each one's delivered frame against
    the frame rank 0 rendered alone, and leaves the fastest that passed AND makes the RCCL call behind every frame up
    for the headline.  A combination that fails on any rank is reported and left out on all of them, not fatal.
    Returns the dict that goes into config.mode_sweep."""
    import numpy as np
    hip, dist, torch, rank, mode = R.hip, R.dist, R.torch, R.rank, R.mode
    first_row, nb_rows = strip

    def configure(per_flight, route, collective):
        """(every rank) tear down what is up, bring this combination up, re-cut the strips; None or why not"""
        if mode["per_flight"] != per_flight:
            loop.sync()
            loop.barrier()
            hip.solr_hip_image_unshare()
            hip.solr_hip_comm_finalize()
            hip.solr_hip_set_strip(first_row, nb_rows)
            mode["delivery"] = None
            if not R.comm_up(per_flight):
                return "the communicator did not come up"
            if balanced and hip.solr_hip_balance_strips() != 0:
                why = R.engine_error()
                hip.solr_hip_clear_error()
                return "solr_hip_balance_strips: " + why
        if mode["delivery"] != route:
            loop.sync()
            got, why = R.delivery_up(route)
            if got != route:
                return why
        mode["collective"] = collective
        for _ in range(PREROLL_FRAMES // 2):
            loop.step()
        loop.drain()
        return None

    sweep = {}
    for per_flight, route, collective in combos:
        name = label(per_flight, route, collective)
        arm("mode sweep: " + name)
        entry = {}
        try:
            why = configure(per_flight, route, collective)
            # every rank takes the same way from here: a combination that did not come up on ONE rank is skipped
            # on all of them (the others would wait in timed()'s barriers for a rank that never came)
            if R.agree(why is not None):
                why = why or "did not come up on another rank"
            if why:
                entry["skipped"] = why
            else:
                seg = loop.timed(steps, warmup, max(regions // 5, 5))
                t = torch.tensor([median(seg["regions"])], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                entry["ms_per_step"] = round(float(t[0]) / steps * 1e3, 5)
                same = True
                if rank == 0 and alone is not None:
                    same = seg["last_image"] is not None and bool(np.array_equal(seg["last_image"], alone))
                entry["frame_equals_single_gpu"] = not R.agree(not same)
                entry["rccl_communicators"] = int(hip.solr_hip_comm_count())
                entry["rccl_calls_per_frame"] = 1 if collective else 0
        except engine_failure as e:            # the engine's error state: reported, cleared, the job goes on
            entry["error"] = str(e)[:300]
        if R.agree("error" in entry):
            entry.setdefault("error", "on another rank")
            hip.solr_hip_clear_error()
            loop.tickets.clear()
        sweep[name] = entry
    usable = {n: e for n, e in sweep.items() if "ms_per_step" in e and e.get("frame_equals_single_gpu") and "error" not in e}
    if not usable:
        raise SystemExit("bench.py rank %d: no communicator mode / delivery route delivered the one-GPU frame: %s"
                         % (rank, json.dumps(sweep)))
    # The headline runs on the route BASELINE.json's north_star names - "a single RCCL gather over xGMI to assemble the
    # final image": the fastest combination that makes that RCCL call behind every frame.  The fastest of ALL of them
    # (which may be the shared host image with no collective per frame) is named beside it, never `value`.
    fastest = min(usable, key=lambda n: usable[n]["ms_per_step"])
    with_rccl = {n: e for n, e in usable.items() if e.get("rccl_calls_per_frame") == 1}
    if not with_rccl:
        raise SystemExit("bench.py rank %d: no combination with the RCCL gather behind every frame delivered the "
                         "one-GPU frame: %s" % (rank, json.dumps(sweep)))
    best = min(with_rccl, key=lambda n: with_rccl[n]["ms_per_step"])
    for per_flight, route, collective in combos:
        if label(per_flight, route, collective) == best:
            arm("mode sweep: back to " + best)
            why = configure(per_flight, route, collective)
            if R.agree(why is not None):
                raise SystemExit("bench.py rank %d: %s did not come up a second time: %s" % (rank, best, why))
    sweep["headline_runs_on"] = best
    sweep["fastest_combination"] = fastest
    return sweep
