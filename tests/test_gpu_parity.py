"""GPU parity tests: the HIP path (through host -> C-ABI -> kernels) against the CPU
oracle on the same flattened scene.  Tolerance (BASELINE.json north_star): primitive ids
and RGB8 exact, float colour channels <= 1 ULP."""
import numpy as np
import pytest

from helpers import compare_frames

pytestmark = pytest.mark.gpu


def render_both(solr, oracle, build, **kw):
    k = solr.Kernel(engine="hip")
    build(k, **kw)
    rgb = k.render()
    pp = k.postprocessing_buffer()
    ids = k.primitive_ids()
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    opp, oids, orgb, counts, status = oracle.render(flat, si, ppi, eye, direction, angles)
    assert status == 0
    k.finalize()
    return compare_frames(pp, ids, rgb, opp, oids, orgb), counts


@pytest.mark.parametrize("iterations", [1, 3])
def test_cornell_parity(solr, oracle, iterations):
    res, counts = render_both(solr, oracle, solr.scenes.cornell, width=160, height=120, iterations=iterations)
    print(res, counts)
    assert res["ids_all_equal"]
    assert res["max_ulp"] <= 1
    assert res["depth_max_ulp"] == 0
    assert res["rgb_equal"]
