"""Dialect 0 of the oracle - the CUDA engine's statements, what the product is held to - pinned by a second,
independent reading of the CUDA text.

oracle/solr_oracle.c is bit for bit the reference's OpenCL engine wherever the two engines agree
(tests/test_reference_probes.py).  The 36 places where they do not are `DIALECT(site)` switches; the CUDA engine
cannot be built in this image, so its arm of every switch used to rest on one reading.  Here:

* tests/cuda_text_model.py is a model of the CUDA path written from /root/reference's .cu / .cuh text alone (plain
  Python over binary32 scalars, every function citing its lines), and tests/cuda_text_cases.py renders some fifty small frames
  and pass sequences with both - every primitive type, both walks, shader, bounce loop with deferred-reflection and
  global-illumination rays, seven cameras, five post-processing kernels, accumulation passes - plus five single-function
  cases whose expected values are worked out from the text in the case itself: frame buffer, ids and bitmap must be
  the SAME BITS;
* a build of the oracle with a counter on every switch (make -C oracle coverage) shows that the cases evaluate the
  CUDA arm of all 36 switches, and no OpenCL arm;
* every switch in turn is made to read the other dialect, alone: some case must then differ from the model.  A switch
  whose two arms no case can tell apart would not be pinned by any of this.  One is known to be unobservable in
  dialect 0 and is asserted to be the only one: the initial value of launchRayTracing's closestPrimitive (CRT:81: -1),
  which the CUDA flow never reads before a hit writes it (the OpenCL engine's depth bookkeeping does).

The cases run in one child process (the counting build must be the first oracle library a process loads)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cuda_text_cases  # noqa: E402

FUNCTION_CASES = ["texture maps on a plane (TM:30-73, 385-441)", "normal map on a plane's own normal (GI:556, TM:37-39)",
                  "camera plane from both sides (GI:514-541, 551-559)", "a textured cone is mapped like a cylinder (GS:46-58)",
                  "a triangle's bump map leaves the opacity alone (TM:260-276)"]
UNOBSERVABLE_IN_DIALECT_0 = {22}     # closestPrimitive's initial value (see the module docstring)


@pytest.fixture(scope="module")
def results():
    res = subprocess.run([sys.executable, os.path.join(HERE, "cuda_text_cases.py")], capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-4000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("name", list(cuda_text_cases.CASES) + FUNCTION_CASES)
def test_oracle_dialect_0_equals_the_cuda_text_model(results, name):
    case = results["cases"][name]
    assert case["same"], json.dumps(case, indent=1)[:3000]
    for p in case.get("passes", []):
        assert p["status"] == 0 and p["frame_buffer"] and p["ids"] and p["bitmap"]


def test_the_frames_are_not_empty(results):
    """most cases fill the frame; none of the scene cases may compare nothing"""
    for name, case in results["cases"].items():
        if "volume camera" in name:      # (its ids are -1 whatever it met, CRT:59)
            assert max(p["lit_pixels"] for p in case["passes"]) >= 150, name
        elif "passes" in case and "box-debug" not in name:
            assert max(p["hit_pixels"] for p in case["passes"]) > 0, name
    filled = sum(1 for c in results["cases"].values() if "passes" in c and max(p["hit_pixels"] for p in c["passes"]) >= 190)
    assert filled >= 35


def test_every_dialect_switch_is_evaluated_in_its_cuda_form(results):
    hits = results["site_hits"]
    assert len(hits) == 36, "a dialect switch was added or removed: revisit the cases (and DESIGN.md section 2)"
    assert all(h > 0 for h in hits), [i for i, h in enumerate(hits) if h == 0]


def test_flipping_any_single_switch_is_noticed(results):
    unnoticed = {f["site"] for f in results["flipped"] if not f["noticed_by"]}
    assert unnoticed == UNOBSERVABLE_IN_DIALECT_0, [f for f in results["flipped"] if not f["noticed_by"]]
