"""A SMOKE ALARM, NOT THE PIN: the oracle against golden frames rendered by THE REFERENCE'S OWN renderer - runs on CPU.

What pins the oracle to reference output is tests/test_reference_probes.py (the reference's own functions over
arrays of inputs, bit for bit; tests/golden/reference_probes.npz), and what holds the ENGINE to reference output is
tests/test_engine_probes_gpu.py.  This file only compares whole frames of the reference's renderer AS BUILT - fused
dot products, approximate reciprocal square roots, the OpenCL engine's own statements where the two engines differ -
with the oracle's CUDA dialect, statistically (the fractions below): it would notice an oracle or a builder that went
badly wrong, and nothing finer.

tests/golden/reference_*.npz hold the outputs of the reference's OpenCL k_standardRenderer + k_default
(compiled for gfx950 from the reference tree, run on an MI355X by tests/golden/make_reference_fixtures.py)
for three small plane-free scenes.  The scene arrays are rebuilt here through the same builder and
checked against the digest stored with the fixture, then the oracle renders the frame and is compared
with the reference's: same tolerances, and the same explanation of the residue (an older sibling engine:
float4 arithmetic with fused dot products, a sub-pixel jitter that is compensated, silhouette pixels), as
tests/test_reference_opencl.py, which repeats the comparison live on the GPU and adds the HIP engine."""
import importlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_reference_fixtures as G  # noqa: E402

# fixture, minimum fraction of pixels: same primitive, identical RGB8, RGB8 within 8 levels, float colour within 1e-5
CASES = [
    ("spheres_1_bounce", 0.9999, 0.985, 0.99, 0.96),       # measured: 100 %, 99.04 %, 99.61 %, 96.8 %
    ("spheres_3_bounces", 0.9999, 0.98, 0.99, 0.955),      # measured: 100 %, 98.57 %, 99.69 %, 96.4 %
    ("triangle_mesh_2_bounces", 0.9999, 0.985, 0.99, 0.98),  # measured: 100 %, 99.17 %, 99.46 %, 98.96 %
]


@pytest.mark.parametrize("name,min_ids,min_rgb,min_rgb8,min_colour", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_the_reference_frames(solr, oracle, name, min_ids, min_rgb, min_rgb8, min_colour):
    golden = np.load(os.path.join(HERE, "golden", "reference_%s.npz" % name))
    k = G.build(solr, G.FIXTURES[name])
    flat = k.flat_scene()
    assert G.scene_digest(flat) == str(golden["scene_digest"]), "the builder no longer produces the scene of the fixture"
    si, ppi, eye, direction, angles = k.frame_parameters()
    pp, ids, rgb, _, status = oracle.render(flat, si, ppi, eye, direction, angles, nthreads=4)
    assert status == 0
    same = float((ids[..., 0] == golden["ids"]).mean())
    diff = np.abs(rgb.astype(int) - golden["rgb"].astype(int)).max(axis=2)
    rel = np.abs(pp[..., :3] - golden["colour"][..., :3]).max(axis=2) / np.maximum(np.abs(pp[..., :3]).max(axis=2), 1e-3)
    res = {"ids_equal": same, "rgb_identical": float((diff == 0).mean()), "rgb_within_8": float((diff <= 8).mean()),
           "colour_within_1e-5": float((rel <= 1e-5).mean()), "colour_median_rel": float(np.median(rel))}
    assert res["ids_equal"] >= min_ids, res
    assert res["rgb_identical"] >= min_rgb, res
    assert res["rgb_within_8"] >= min_rgb8, res
    assert res["colour_within_1e-5"] >= min_colour, res
    assert res["colour_median_rel"] <= 1e-5, res
