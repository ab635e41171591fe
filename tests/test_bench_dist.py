"""(CPU) bench_dist.mode_sweep and Ranks with a stand-in for the engine library and a one-rank control plane: which
combination the headline of an N > 1 job runs on (the fastest that delivers the one-GPU frame AND makes the RCCL call behind
every frame - never the route without a collective, however fast), what is reported beside it, and that a combination
which fails to come up, delivers a wrong frame or raises inside its timed segment is reported and left out, not fatal."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_dist  # noqa: E402


class FakeDist:
    class ReduceOp:
        MAX, SUM = "max", "sum"

    def all_reduce(self, t, op=None):
        return None

    def barrier(self):
        return None

    def broadcast_object_list(self, box, src=0):
        return None


class EngineFailure(Exception):
    pass


class FakeHip:
    """the calls bench_dist makes, with a communicator that can be told to fail in one mode"""

    def __init__(self, fail_per_flight=False, fail_share=False):
        self.per_flight, self.comms, self.shared = 0, 0, False
        self.fail_per_flight, self.fail_share = fail_per_flight, fail_share
        self.log = []

    def solr_hip_comm_set_per_flight(self, v):
        self.per_flight = v

    def solr_hip_comm_unique_id(self, uid):
        return 0

    def solr_hip_comm_init(self, rank, world, uid):
        if self.per_flight and self.fail_per_flight:
            return -1
        self.comms = 4 if self.per_flight else 1
        return 0

    def solr_hip_comm_count(self):
        return self.comms

    def solr_hip_comm_ranks(self):
        return 1 if self.comms else 0

    def solr_hip_comm_finalize(self):
        self.comms = 0

    def solr_hip_clear_error(self):
        self.log.append("clear")

    def solr_hip_last_error(self, buf, n):
        return 0

    def solr_hip_image_unshare(self):
        self.shared = False

    def solr_hip_image_share(self, name, rank, world):
        if self.fail_share:
            return -1
        self.shared = True
        return 0

    def solr_hip_image_share_sealed(self):
        return None

    def solr_hip_set_strip(self, first, rows):
        return None

    def solr_hip_balance_strips(self):
        return 0


def _ranks(hip):
    import torch
    return bench_dist.Ranks(FakeDist(), torch, hip, 0, 1, collective=True)


def _loop(R, times, wrong=(), raises=()):
    """a loop whose timed segment takes what `times` says for the combination that is up"""
    alone = np.arange(12, dtype=np.uint8).reshape(2, 2, 3)
    state = {"steps": 0}

    def name():
        return bench_dist.label(R.mode["per_flight"], R.mode["delivery"], R.mode["collective"])

    def timed(steps, warmup, regions):
        n = name()
        if n in raises:
            raise EngineFailure("the engine gave up in " + n)
        image = alone.copy()
        if n in wrong:
            image[0, 0, 0] ^= 1
        return {"regions": [times[n] * steps * 1e-3] * regions, "last_image": image}

    def step():
        state["steps"] += 1

    loop = types.SimpleNamespace(step=step, drain=lambda: None, sync=lambda: None, barrier=lambda: None, timed=timed,
                                 tickets=[1, 2, 3])
    return loop, alone, state


COMBOS = [(False, "strips", True), (False, "gathered", True), (True, "strips", True), (True, "gathered", True),
          (False, "strips", False)]
NAMES = [bench_dist.label(*c) for c in COMBOS]


def _sweep(R, loop, alone, combos=COMBOS):
    return bench_dist.mode_sweep(R, loop, combos, steps=10, warmup=1, regions=25, alone=alone, balanced=True, strip=(0, 8),
                                 arm=lambda phase: None, engine_failure=EngineFailure)


def test_the_headline_is_the_fastest_combination_that_gathers_with_rccl():
    hip = FakeHip()
    R = _ranks(hip)
    times = dict(zip(NAMES, [0.050, 0.060, 0.045, 0.070, 0.030]))   # the route without a collective is the fastest of all
    loop, alone, _ = _loop(R, times)
    sweep = _sweep(R, loop, alone)
    assert sweep["fastest_combination"] == "no_collective_strips_over_every_ranks_link"
    assert sweep["headline_runs_on"] == "communicator_per_flight_strips_over_every_ranks_link"
    assert sweep[sweep["headline_runs_on"]]["rccl_calls_per_frame"] == 1
    # ... and that combination is what is up when the sweep returns
    assert R.mode == {"per_flight": True, "delivery": "strips", "collective": True} and hip.comms == 4 and hip.shared
    assert all(sweep[n]["frame_equals_single_gpu"] for n in NAMES)
    assert [sweep[n]["ms_per_step"] for n in NAMES] == [0.05, 0.06, 0.045, 0.07, 0.03]
    assert R.require_communicator() == (1, 4)


def test_a_combination_that_does_not_come_up_is_skipped_and_the_next_one_starts_from_nothing():
    """after a communicator mode that fails, nothing is up - the next combination must bring one up
    again whatever it asks for"""
    hip = FakeHip(fail_per_flight=True)
    R = _ranks(hip)
    times = dict(zip(NAMES, [0.050, 0.040, 0.045, 0.070, 0.030]))
    loop, alone, _ = _loop(R, times)
    combos = [COMBOS[2], COMBOS[0], COMBOS[1], COMBOS[4]]           # the failing mode FIRST
    sweep = _sweep(R, loop, alone, combos)
    assert "skipped" in sweep["communicator_per_flight_strips_over_every_ranks_link"]
    assert sweep["headline_runs_on"] == "one_communicator_gathered_frame_over_rank0s_link"
    assert hip.comms == 1 and R.mode["per_flight"] is False and R.mode["delivery"] == "gathered"


def test_a_wrong_frame_or_an_engine_error_leaves_a_combination_out_not_the_job():
    hip = FakeHip()
    R = _ranks(hip)
    times = dict(zip(NAMES, [0.020, 0.060, 0.045, 0.070, 0.030]))
    loop, alone, _ = _loop(R, times, wrong={NAMES[0]}, raises={NAMES[2]})
    sweep = _sweep(R, loop, alone)
    assert sweep[NAMES[0]]["frame_equals_single_gpu"] is False      # the fastest of all delivers a wrong frame
    assert "error" in sweep[NAMES[2]] and loop.tickets == []        # the engine's error: reported, cleared, tickets dropped
    assert sweep["headline_runs_on"] == NAMES[1] and sweep["fastest_combination"] == NAMES[4]


def test_no_combination_with_the_gather_left_voids_the_job():
    hip = FakeHip()
    R = _ranks(hip)
    times = dict(zip(NAMES, [0.05] * 5))
    loop, alone, _ = _loop(R, times, wrong=set(NAMES[:4]))
    with pytest.raises(SystemExit) as e:
        _sweep(R, loop, alone)
    assert "RCCL gather behind every frame" in str(e.value)


def test_a_box_that_cannot_share_the_host_image_falls_back_to_the_gathered_route_with_the_gather_on():
    """the fall-back route delivers what the gather assembled - no gather, no frame"""
    hip = FakeHip(fail_share=True)
    R = _ranks(hip)
    R.mode["collective"] = False
    assert R.comm_up(False)
    route, why = R.delivery_up("strips")
    assert route == "gathered" and "solr_hip_image_share failed" in why and R.mode["collective"] is True


def test_a_communicator_of_another_size_voids_the_job():
    hip = FakeHip()
    import torch
    R = bench_dist.Ranks(FakeDist(), torch, hip, 0, 2, collective=True)   # --gpus 2, a communicator of one rank
    assert R.comm_up(False)
    with pytest.raises(SystemExit) as e:
        R.require_communicator()
    assert e.value.code == 5
