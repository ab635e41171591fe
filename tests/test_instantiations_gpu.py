"""Every instantiation of the renderer kernel on the same scene: the host launches the smallest one that covers a
scene's primitive types, textures and camera (solr_launch.hip, the `variants` ladder), so a plain scene only ever runs
the lean ones.  SOLR_HIP_FORCE_FEATURES=mask (read once per process) makes the engine choose as if the scene had
those features too: the Cornell box and a molecule through each step of the ladder must be the frame the oracle AS
PINNED renders - ids exact, RGB8 exact, float colour <= 1 ULP but for at most two pixels behind a mis-rounded powf
(helpers.assert_frame_pinned) - and the same bits as through their own kernel."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = r"""
import importlib, json, os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
solr = importlib.import_module("sol-r_amd")
from oracle import loader
from helpers import assert_frame_pinned, gpu_frame
out = {}
# (ten bounces: the lean rows answer with their F_STACK instantiation - three colour-stack slots in LDS, the deeper ones
# in HBM - the other rows with the whole stack in LDS)
for name, build, kw in (("cornell", solr.scenes.cornell, dict(width=96, height=64, iterations=3)),
                        ("cornell, ten bounces", solr.scenes.cornell, dict(width=96, height=64, iterations=10)),
                        ("molecule", solr.scenes.molecule, dict(atoms=400, width=96, height=64))):
    k = solr.Kernel(engine="hip")
    build(k, **kw)
    pp, ids, rgb = gpu_frame(k)
    # the oracle as pinned: every pixel at the bar but for at most 2 that went through a mis-rounded powf (helpers)
    res = assert_frame_pinned(k, loader, (pp, ids, rgb), 2, name)
    res["status"] = 0
    res["digest"] = [int(pp.view(np.uint32).sum(dtype=np.uint64)), int(ids.astype(np.int64).sum()), int(rgb.astype(np.int64).sum())]
    out[name] = res
    k.finalize()
print(json.dumps(out))
"""

# rt_device.h enum Feature: SPHERE 1, PROC 2, CYL 4, ELL 8, TRI 16, PLANE 32, TEX 64, FULL 128
LADDER = {"the scene's own": 0, "sphere + plane + triangle + cylinder": 53, "textured sphere + triangle": 81, "the textured mix": 117,
          "the untextured mix with the special cameras": 181, "every type + textures": 127, "everything": 255}


def _run(mask):
    env = dict(os.environ, SOLR_HIP_FORCE_FEATURES=str(mask))
    code = CHILD % {"root": os.path.dirname(HERE), "here": HERE}
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def own():
    return _run(0)


@pytest.mark.parametrize("name", list(LADDER))
def test_every_instantiation_renders_the_oracles_frame(own, name):
    got = own if LADDER[name] == 0 else _run(LADDER[name])
    for scene, res in got.items():
        assert res["status"] == 0 and res["ids_all_equal"] and res["depth_max_ulp"] == 0, (scene, res)
        assert res["pixels_outside_the_bar"] <= 2 and res["max_ulp"] <= 2 and res["rgb_max_diff"] <= 1, (scene, res)
        assert res["digest"] == own[scene]["digest"], (scene, "not the bits of the scene's own kernel")
