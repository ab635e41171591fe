"""The N > 1 frame loop of bench.py (sol-r_amd.StripPipeline) on one GPU: an RCCL process group of one
rank, strips gathered to rank 0, one and two frames in flight (tests/strip_pipeline_worker.py).  Runs in a
process of its own because torch must be imported before the engine library initialises the HIP
runtime, and the other GPU tests of this session have long done that."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("flights", [3, 2, 1])
def test_gathered_frames_are_the_rendered_frames(flights):
    res = subprocess.run([sys.executable, os.path.join(HERE, "strip_pipeline_worker.py"), str(flights)],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "STRIP_PIPELINE_OK %d" % flights in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]
