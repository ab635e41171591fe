"""bench.py at N > 1.  The driver starts it bare (`python bench.py --gpus N`): it must then launch its own rank
processes, relay rank 0's JSON line, and fail cleanly - no hang, a non-zero exit - when a rank cannot run.

CPU: the launcher in this GPU-less container (both ranks get as far as "needs a GPU").  GPU: a rehearsal of the whole
N = 2 job on the box's one GPU - SOLR_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and tests/loopback_rccl.c stands in
for RCCL (which refuses two ranks on one device) - so that the strips, the communicator, the balance, both timed
segments, the gather-alone timing and the gathered-frame check of the JSON line have all run with two ranks before
the first real 8-GPU run.  Its numbers mean nothing; its `gathered_equals_single_gpu` does."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_launch_without_gpus_fails_cleanly(solr, have_gpu):
    if have_gpu:
        pytest.skip("this is the GPU-less container's test")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=180)
    assert res.returncode != 0
    assert "rank 0 needs a GPU" in res.stderr and "rank 1 needs a GPU" in res.stderr, res.stderr[-2000:]
    assert "the job is void" in res.stderr
    assert res.stdout.strip() == ""          # no JSON line from a job that did not run


def test_a_launchers_environment_is_respected(solr, have_gpu):
    """under torch.distributed.run (RANK / WORLD_SIZE set) bench.py is a rank, not a launcher; a world size that
    does not match --gpus is refused at once"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert res.returncode == 2 and "WORLD_SIZE=1" in res.stderr


def test_a_wrong_frame_or_a_wrong_ray_count_is_seen(solr, oracle):
    """(CPU) bench.prove_frames - what bench.py runs on the image its last timed step delivered: the oracle's own frame
    passes, a stub engine's frame with sixteen wrong pixels does not, nor does a census that is one ray off"""
    import importlib
    import numpy as np
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    k = solr.Kernel(engine="host-only", deterministic_seed=1)
    solr.scenes.cornell(k, width=96, height=64, iterations=3)
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    _, _, orgb, counts, status = oracle.render(flat, si, ppi, eye, direction, angles)
    k.finalize()
    rays = int(counts[0]) + int(counts[1])
    assert status == 0 and rays > 96 * 64
    good = bench.prove_frames(flat, si, ppi, eye, direction, angles, orgb, rays)
    assert good["delivered_frame_equals_oracle"] is True and good["rays_equal_oracle_count"] is True
    assert good["oracle_rays_per_frame"] == rays and good["frame_check"]["pixels_differing"] == 0
    wrong = orgb.copy()
    wrong[32, 40:56] ^= 0x40
    bad = bench.prove_frames(flat, si, ppi, eye, direction, angles, wrong, rays + 1)
    assert bad["delivered_frame_equals_oracle"] is False and bad["rays_equal_oracle_count"] is False
    assert bad["frame_check"]["pixels_differing"] == 16
    # one RGB8 step on a pixel the oracle did not mark is a wrong frame too
    nudged = orgb.copy()
    nudged[10, 10, 0] = nudged[10, 10, 0] + 1 if nudged[10, 10, 0] < 255 else 254
    assert bench.prove_frames(flat, si, ppi, eye, direction, angles, nudged, rays)["delivered_frame_equals_oracle"] is False


def _rehearse(extra, timeout, **more_env):
    from test_multi_rank_gpu import build_loopback
    directory = tempfile.mkdtemp(prefix="solr_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        env.update(SOLR_BENCH_SHARE_GPU="1", SOLR_HIP_RCCL_LIBRARY=build_loopback(), SOLR_LOOPBACK_DIR=directory,
                   SOLR_LOOPBACK_TIMEOUT="60", SOLR_BENCH_TIMEOUT=str(timeout), SOLR_BENCH_REGIONS="3")
        env.update(more_env)     # (three timed regions: a rehearsal's numbers mean nothing, and its transport is files)
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"] + extra,
                             env=env, capture_output=True, text=True, timeout=timeout + 60)
        assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
        return json.loads(res.stdout.strip().splitlines()[-1])
    finally:
        shutil.rmtree(directory, ignore_errors=True)


@pytest.mark.gpu
def test_two_rank_rehearsal_of_the_default_job(solr):
    line = _rehearse(["--steps", "12", "--warmup", "3", "--width", "640", "--height", "360"], 600)
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["steps"] == 12 and line["warmup"] == 3 and line["scaling"] == "strong"
    assert cfg["rccl_ranks"] == 2 and cfg["strips"] == "balanced by cost"
    assert cfg["gathered_equals_single_gpu"] is True
    assert len(cfg["per_rank"]) == 2 and [r["rank"] for r in cfg["per_rank"]] == [0, 1]
    rows = [r["rows"] for r in cfg["per_rank"]]
    assert rows[0][0] == 0 and rows[0][1] + rows[1][1] == 360 and rows[1][0] == rows[0][1] and rows[0][1] % 8 == 0
    assert [r["equal_strips"]["rows"] for r in cfg["per_rank"]] == [[0, 180], [180, 180]]
    assert cfg["slowest_rank"] in (0, 1) and cfg["gather_only_ms"] > 0
    assert {"balanced_strips_native_gather", "equal_strips_native_gather"} <= set(cfg["rates_mrays_per_s"])
    assert cfg["rates_mrays_per_s"]["balanced_strips_native_gather"] == pytest.approx(line["value"], rel=1e-3)
    # the default job times both communicator modes and both delivery routes - and the strips route with no RCCL call per
    # frame at all - in short segments, checks each one's frame against the one-GPU frame, and runs the headline on the
    # fastest that passed AND gathers with RCCL behind every frame (the route BASELINE.json's north_star names); the
    # fastest of all of them is named beside it
    sweep = cfg["mode_sweep"]
    combos = [n for n in sweep if n not in ("headline_runs_on", "fastest_combination")]
    assert "no_collective_strips_over_every_ranks_link" in combos and sweep["no_collective_strips_over_every_ranks_link"]["rccl_calls_per_frame"] == 0
    assert len(combos) == 5 and all(sweep[n]["frame_equals_single_gpu"] is True and sweep[n]["ms_per_step"] > 0 for n in combos)
    with_rccl = [n for n in combos if sweep[n]["rccl_calls_per_frame"] == 1]
    assert len(with_rccl) == 4
    assert sweep["headline_runs_on"] == min(with_rccl, key=lambda n: sweep[n]["ms_per_step"])
    assert sweep["fastest_combination"] == min(combos, key=lambda n: sweep[n]["ms_per_step"])
    assert line["protocol"] == "delivered_pipelined_behind_the_rccl_gather"
    assert {sweep[n]["rccl_communicators"] for n in combos} == {1, 4}
    assert all("sweep_" + n in cfg["rates_mrays_per_s"] for n in combos)
    assert "rehearsal" in cfg and line["roofline"]["frac"] > 0 and "cpu_baseline" not in line
    assert cfg["step_ms_spread"]["min"] <= cfg["step_ms_spread"]["median"] <= cfg["step_ms_spread"]["max"]
    # ms_per_step IS the median region, the spread its fastest and slowest: it brackets the headline by construction
    assert line["regions"] == 3 == cfg["step_ms_spread"]["regions"]
    assert cfg["step_ms_spread"]["median"] == pytest.approx(line["ms_per_step"], rel=1e-3)
    # every frame of the timed regions was delivered to rank 0's host memory
    assert cfg["frames_delivered"] >= 3 * 12
    if "strips" in sweep["headline_runs_on"]:
        assert "one image the ranks' processes share" in cfg["delivery"]
    else:
        assert "gather on rank 0" in cfg["delivery"]
    assert ("none in the data path" in cfg["gather"]) == sweep["headline_runs_on"].startswith("no_collective")
    assert cfg["rccl_communicators"] == (4 if "per_flight" in sweep["headline_runs_on"] else 1)


@pytest.mark.gpu
def test_two_rank_rehearsal_with_one_communicator_per_flight(solr):
    """the switch the first real N > 1 run can A/B (RCCL orders the operations of one communicator across streams)"""
    line = _rehearse(["--steps", "12", "--warmup", "3", "--width", "640", "--height", "360"], 600,
                     SOLR_HIP_COMM_PER_FLIGHT="1")
    cfg = line["config"]
    assert cfg["rccl_ranks"] == 2 and cfg["rccl_communicators"] == 4 and "one per frame in flight" in cfg["rccl_communicator_mode"]
    assert cfg["gathered_equals_single_gpu"] is True
    # the environment fixed the communicator mode: only the delivery routes were swept
    assert len([n for n in cfg["mode_sweep"] if n not in ("headline_runs_on", "fastest_combination")]) == 3
    assert all("communicator_per_flight" in n or n.startswith("no_collective") for n in cfg["mode_sweep"] if n not in ("headline_runs_on", "fastest_combination"))


@pytest.mark.gpu
def test_two_rank_rehearsal_delivering_the_gathered_frame(solr):
    """--delivery gathered: rank 0 copies the frame the gather assembled in its HBM (the A/B of the delivery route)"""
    line = _rehearse(["--steps", "12", "--warmup", "3", "--width", "640", "--height", "360", "--delivery", "gathered"], 600)
    cfg = line["config"]
    assert "gather on rank 0" in cfg["delivery"] and cfg["gathered_equals_single_gpu"] is True
    assert cfg["frames_delivered"] >= 3 * 12


@pytest.mark.gpu
def test_a_box_that_cannot_share_the_host_image_falls_back_to_the_gathered_frame(solr):
    """one rank cannot open the shared segment: every rank falls back, together, to rank 0 copying the gathered frame -
    the job is not void, and its line says what happened"""
    line = _rehearse(["--steps", "12", "--warmup", "3", "--width", "640", "--height", "360"], 600, SOLR_BENCH_FAIL_SHARE="1")
    cfg = line["config"]
    assert "solr_hip_image_share failed" in cfg["delivery_fallback"] and "gather on rank 0" in cfg["delivery"]
    assert cfg["gathered_equals_single_gpu"] is True and cfg["frames_delivered"] >= 3 * 12
    # the routes that wanted the shared image say why they were left out; the job ran on a gathered one
    sweep = cfg["mode_sweep"]
    assert "gathered" in sweep["headline_runs_on"]
    assert all("skipped" in sweep[n] for n in sweep if "strips" in n and n not in ("headline_runs_on", "fastest_combination"))


@pytest.mark.gpu
def test_two_rank_rehearsal_of_cfg4(solr):
    """BASELINE configs[4] on two ranks: passes 0...73 at 3840 x 2160, natural depth of field, ambient occlusion
    through the depth-halo exchange on cost-balanced strips; the assembled frames after passes 0, 11 and 73 equal
    the ones rank 0 renders alone"""
    line = _rehearse(["--config", "cfg4", "--steps", "74", "--warmup", "2"], 1200)
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["gathered_equals_single_gpu"] is True and cfg["rccl_ranks"] == 2
    rows = [r["rows"] for r in cfg["per_rank"]]
    assert rows[0][1] + rows[1][1] == 2160


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_the_whole_job_over_real_rccl_where_the_box_has_the_gpus(solr, world):
    """`python bench.py --gpus N` as the driver starts it, on N REAL GPUs over the real RCCL - runs by itself on the first
    box that has them (the test boxes have one GPU: skipped): the five combinations of the mode sweep, the frame of
    every one of them equal to the one-GPU frame, the headline on the fastest"""
    if solr.hip_lib().solr_hip_device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SOLR_BENCH_SHARE_GPU",
                                                            "SOLR_HIP_RCCL_LIBRARY")}
    env.update(SOLR_BENCH_REGIONS="5", MASTER_PORT=str(29650 + world))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    line = json.loads(res.stdout.strip().splitlines()[-1])
    cfg = line["config"]
    assert line["n_gpus"] == world and cfg["rccl_ranks"] == world and cfg["gathered_equals_single_gpu"] is True
    assert cfg["delivered_frame_equals_oracle"] is True and cfg["rays_equal_oracle_count"] is True
    sweep = cfg["mode_sweep"]
    assert all(sweep[n].get("frame_equals_single_gpu") is True for n in sweep if n not in ("headline_runs_on", "fastest_combination")), sweep
    assert "rehearsal" not in cfg


@pytest.mark.gpu
def test_one_rank_through_the_distributed_path_over_rccl(solr):
    """SOLR_BENCH_FORCE_DIST=1: the N > 1 path with a communicator of one rank of the real RCCL"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(SOLR_BENCH_FORCE_DIST="1", MASTER_PORT="29613")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--no-cpu-baseline",
                          "--width", "640", "--height", "360"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1 and line["config"]["gathered_equals_single_gpu"] is True
    assert line["config"]["per_rank"][0]["rows"] == [0, 360]


@pytest.mark.gpu
def test_the_drivers_own_command_line_for_two_ranks(solr):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py
    --gpus 2 --steps K --warmup W` - how the driver starts an N > 1 run - rehearsed on the box's one GPU (every rank on
    GPU 0, the stand-in transport): one JSON line on stdout, last, from rank 0"""
    from test_multi_rank_gpu import build_loopback
    directory = tempfile.mkdtemp(prefix="solr_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
        env.update(SOLR_BENCH_SHARE_GPU="1", SOLR_HIP_RCCL_LIBRARY=build_loopback(), SOLR_LOOPBACK_DIR=directory,
                   SOLR_LOOPBACK_TIMEOUT="60", SOLR_BENCH_REGIONS="3", HSA_ENABLE_IPC_MODE_LEGACY="0")
        res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "bench.py"),
                              "--gpus", "2", "--steps", "10", "--warmup", "2", "--width", "640", "--height", "360"],
                             env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
        lines = [l for l in res.stdout.strip().splitlines() if l.strip()]
        line = json.loads(lines[-1])
        assert line["n_gpus"] == 2 and line["steps"] == 10 and line["warmup"] == 2 and line["config"]["rccl_ranks"] == 2
        assert line["config"]["gathered_equals_single_gpu"] is True and line["value"] > 0
        assert "cpu_baseline" not in line            # (N > 1: the CPU baseline is rank 0's at N = 1 only)
    finally:
        shutil.rmtree(directory, ignore_errors=True)


@pytest.mark.gpu
def test_the_one_gpu_line_keeps_the_contract(solr):
    """`python bench.py --gpus 1 --steps K --warmup W`: one JSON line, last on stdout, with the keys the driver reads -
    and the headline inside its own error bar"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--cpu-seconds", "1.5"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    line = json.loads(res.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"].startswith("Mrays/s @1920x1080, 3-bounce Cornell") and line["unit"] == "Mrays/s"
    assert (line["n_gpus"], line["steps"], line["warmup"]) == (1, 20, 5) and line["higher_is_better"] is True
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert "model" not in line["config"] and "workload" in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3) and 0.0 < roof["frac"] < 1.0
    assert roof["traffic"] is None or roof["traffic"] > 0.5 * roof["algorithmic_bytes"]
    # the walk's own ceiling is above the headline, and the node loop alone is a part of the kernel's time
    assert roof["walk_bound_mrays"] > line["value"] and 0.1 < roof["walk_bound"]["frac_of_bound_achieved"] < 1.0
    assert roof["walk_bound"]["node_loop_only_ms"] < roof["kernel_ms"] and roof["walk_bound"]["walks_not_replayed"] == 0
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0
    # value = rays x steps / the median region; the spread is the fastest and the slowest region and brackets it
    sp = line["config"]["step_ms_spread"]
    assert line["regions"] >= 25 and sp["regions"] == line["regions"]
    assert sp["min"] <= line["ms_per_step"] <= sp["max"] and sp["median"] == pytest.approx(line["ms_per_step"], rel=1e-3)
    assert line["value"] == pytest.approx(line["config"]["rays_per_frame"] / (line["ms_per_step"] * 1e-3) / 1e6, rel=2e-3)
    assert sp["timed_seconds_in_all"] >= 0.15                      # the GPU was busy long enough to be seen
    assert line["config"]["frames_delivered"] >= line["regions"] * 20
    # the line proves what it timed: the last delivered image is the oracle's frame, the census the oracle's count
    cfg = line["config"]
    assert cfg["delivered_frame_equals_oracle"] is True and cfg["rays_equal_oracle_count"] is True
    assert cfg["rays_per_frame"] == cfg["oracle_rays_per_frame"] == cfg["engine_census_rays_per_frame"]
    assert cfg["frame_check"]["of_which_not_behind_a_misrounded_libm_result"] == 0


@pytest.mark.gpu
def test_a_job_that_delivers_a_wrong_frame_fails(solr):
    """SOLR_BENCH_TAMPER=1 damages sixteen pixels of the delivered image before the check: the line is still printed,
    says so, and the exit code is not 0"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(SOLR_BENCH_TAMPER="1", SOLR_BENCH_REGIONS="3")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2",
                          "--no-cpu-baseline", "--no-walk-bound", "--width", "640", "--height", "360"], env=env,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 4, (res.returncode, res.stderr[-2000:])
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["config"]["delivered_frame_equals_oracle"] is False and line["config"]["rays_equal_oracle_count"] is True
    assert line["config"]["frame_check"]["pixels_differing"] == 16
