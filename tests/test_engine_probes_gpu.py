"""THE ENGINE against THE REFERENCE'S OWN FUNCTIONS, element for element, with no oracle in between.

tests/golden/reference_probes.npz holds arrays of inputs and what the reference's own functions (RayTracer.cl, wrapped
by oracle/ref_probes.cl, source-order build: oracle/Makefile) returned for them on an MI355X.  The test-only entry
points of include/solr_hip_probes.h evaluate the ENGINE's device functions - the ones k_standardRenderer is built
from, in the instantiations the renderer launches - on the same arrays (tests/engine_probes.py), and the results are
compared with the reference's BIT FOR BIT wherever the function is not under one of the 36 statements in which the
reference's two engines differ (the fixture is the OpenCL engine's, the product is held to the
CUDA engine's):

  function                                   engine == reference, bit for bit, on
  boxIntersection, three forms               every ray without a zero direction component (switch 0: the reciprocal of
    (compare chain, sign-free, node loop)      0); the hand-scheduled node loop on the elements with t0 = 0, its only form
  sphere / ellipsoid / cylinder / cone /     hit flag, hit point, shadow intensity of EVERY element, normal of every
    plane / checkerboard / triangle tests      primitive but triangles (switches 8-9: areas raw, normal normalised); the
    as both walks dispatch them                double-sided rule of triangles is switch 10
  intersectionWithPrimitives                 Cornell room with glass, sticks: every output of every ray; triangle meshes
    (lean and all-features instantiations,     and the textured scene: hit, primitive, hit point of every ray, normal and
    walk-order list / order-free lists /       areas where the primitive is no triangle; the mixed scene (cones: switch 11)
    the reference's own list)                  wherever the two pick the same primitive
  intersectionShader + mappers + maps        colour, bump normal, attributes, ambient occlusion of every element; the
                                               specular vector of everything but spheres (switch 2)
  skyboxMapping, vectorRefraction /          every element
    vectorReflection, makeColor RGB / BGR

Where a switch applies the engine must equal the oracle's CUDA dialect on those elements (and does on ALL elements of
every case: asserted as well) - that is the residue that rests on the reading of the CUDA text
(tests/cuda_text_model.py).  processShadows sits under three switches (14-16: node test from 0, the shaded primitive
left out, the transparent-shadow factor) and is compared with the CUDA dialect only.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import engine_probes as E  # noqa: E402

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_probes.npz")
ptTriangle, ptSphere, ptCone = 2, 0, 12


@pytest.fixture(scope="module")
def probes(solr, oracle):
    from oracle import probes as module
    return module


@pytest.fixture(scope="module")
def fixture():
    return np.load(FIXTURE)


def _load(probes, fixture, name):
    inputs = {k.split("/", 2)[2]: fixture[k] for k in fixture.files if k.startswith(name + "/in/")}
    assert inputs, name
    case = probes.unpack(inputs)
    ref = {k.split("/")[2]: fixture[k] for k in fixture.files if k.startswith(name + "/source_order/")}
    return case, ref


def _rows_equal(probes, a, b):
    same = probes.same_bits(np.asarray(a), np.asarray(b))
    return same.reshape(len(same), -1).all(axis=1)


def _names():
    from oracle import probes as module
    return [n for n in module.CASES if n.split("_")[0] in ("box", "primitive", "closest", "shadow", "vectors", "make",
                                                           "skybox", "intersection")]


def _all_equal(probes, what, mine, theirs, where=None):
    eq = _rows_equal(probes, mine, theirs)
    if where is not None:
        eq = eq | ~where
    assert eq.all(), "%s: %d of %d elements differ (first: %d)" % (what, int((~eq).sum()), len(eq), int(np.flatnonzero(~eq)[0]))


def _variants(case):
    if case["name"] in ("closest", "shadow"):
        return [("the renderer's instantiation, walk-order + order-free lists", 0, 0),
                ("the renderer's instantiation, the reference's own list", 0, 1),
                ("all-features instantiation", E.EVERYTHING, 0)]
    if case["name"] == "primitive":
        return [("the renderer's instantiation", 0, 0), ("all-features instantiation", E.EVERYTHING, 0)]
    return [("", 0, 0)]


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names())
def test_engine_functions_reproduce_the_reference_functions(solr, probes, oracle, fixture, name):
    case, ref = _load(probes, fixture, name)
    cuda = probes._oracle_outputs(oracle.lib(), case)            # the oracle's CUDA dialect: only where a switch applies
    assert oracle.lib().oracle_get_dialect() == 0
    kind = case["name"]
    for label, features, exact in _variants(case):
        out = E.engine_outputs(solr, case, features=features, exact=exact)
        what = "%s (%s)" % (name, label)
        if kind == "box":
            no_zero = (case["directions"] != 0).all(axis=1)
            near0 = case["t0"] == 0
            assert no_zero.sum() > 400 and (no_zero & near0).sum() > 150 and (~no_zero).sum() > 100
            for key in ("hit", "hit_fast"):
                assert (out[key] >= 0).all()                      # every ray of the case meets the sign-free form's precondition
                _all_equal(probes, what + " " + key + " vs the reference", out[key], ref["hit"], no_zero)
                _all_equal(probes, what + " " + key + " vs the CUDA dialect", out[key], cuda["hit"])
            for key in ("hit_walk", "hit_walk_deep"):             # the node loop tests [0, far): its only form in both walks
                _all_equal(probes, what + " " + key + " vs the reference", out[key], ref["hit"], no_zero & near0)
                _all_equal(probes, what + " " + key + " vs the CUDA dialect", out[key], cuda["hit"], near0)
            assert 0.05 < ref["hit"].mean() < 0.95
        elif kind == "primitive":
            types = case["prims"]["type"]
            triangle = (types == ptTriangle) | (case["si"].extendedGeometry == 0)
            # switch 10: with double-sided triangles the CUDA text's dangling else makes every shadow test miss and keeps
            # every other hit (GI:638-648); the OpenCL engine rejects by the side the ray comes from (CL:1379-1394)
            switch10 = triangle & (case["si"].doubleSidedTriangles != 0)
            _all_equal(probes, what + " hit", out["hit"], ref["hit"], ~switch10)
            hit = (out["hit"] != 0) & (ref["hit"] != 0)
            assert hit.sum() > 0.15 * len(hit)
            _all_equal(probes, what + " hit point", out["intersection"], ref["intersection"], hit)
            _all_equal(probes, what + " shadow intensity", out["shadow"], ref["shadow"], hit)
            _all_equal(probes, what + " normal", out["normal"], ref["normal"], hit & ~triangle)
            _all_equal(probes, what + " areas", out["areas"], ref["areas"], hit & ~triangle)
            if case["si"].extendedGeometry:
                assert (hit & ~triangle).sum() > 500
            for key in ("hit", "intersection", "normal", "areas", "shadow"):     # ... and the CUDA dialect everywhere
                _all_equal(probes, what + " %s vs the CUDA dialect" % key, out[key], cuda[key],
                           None if key == "hit" else (out["hit"] != 0) & (cuda["hit"] != 0))
        elif kind == "closest":
            s = case["scene"]
            agree = (out["hit"] == ref["hit"]) & ((out["primitive"] == ref["primitive"]) | (out["hit"] == 0))
            if name == "closest_mix":
                assert agree.mean() > 0.9                          # the cone is not in the OpenCL engine's dispatch (switch 11)
            else:
                assert agree.all(), (what, int((~agree).sum()))
            hit = agree & (out["hit"] != 0)
            assert hit.sum() > 100
            prim_type = s.prims["type"][np.clip(out["primitive"], 0, len(s.prims) - 1)]
            triangle = (prim_type == ptTriangle) | (case["si"].extendedGeometry == 0)
            _all_equal(probes, what + " hit point", out["intersection"], ref["intersection"], hit)
            _all_equal(probes, what + " normal", out["normal"], ref["normal"], hit & ~triangle)
            _all_equal(probes, what + " areas", out["areas"], ref["areas"], hit & ~triangle)
            if name in ("closest_cornell", "closest_sticks"):
                assert not triangle[hit].any()                     # nothing of these two cases is under a switch
            for key in ("hit", "primitive", "intersection", "normal", "areas"):
                _all_equal(probes, what + " %s vs the CUDA dialect" % key, out[key], cuda[key],
                           None if key == "hit" else (out["hit"] != 0) & (cuda["hit"] != 0))
        elif kind == "shadow":
            # three switches in this one function: the CUDA dialect is what there is to compare with.  (The reference's
            # probe leaves out the lamp only; so does this call: engine_probes passes an index nobody has.)
            nobody_cuda = cuda
            _all_equal(probes, what + " result vs the CUDA dialect", out["result"], nobody_cuda["result"])
            _all_equal(probes, what + " colour vs the CUDA dialect", out["color"], nobody_cuda["color"])
            assert 0.05 < (out["result"] > 0).mean() < 0.95
        elif kind == "intersection_shader":
            sphere = case["prims"]["type"] == ptSphere
            for key in ("color", "bump", "advanced", "attributes"):
                _all_equal(probes, what + " " + key, out[key], ref[key])
            _all_equal(probes, what + " specular", out["specular"], ref["specular"], ~sphere)      # switch 2: specularMap
            _all_equal(probes, what + " specular vs the CUDA dialect", out["specular"], cuda["specular"])
            assert (~sphere).sum() > 300
        else:   # skybox, vectors, make_color
            for key, value in out.items():
                if key != "features":
                    _all_equal(probes, what + " " + key, value, ref[key])


@pytest.mark.gpu
def test_the_probes_run_the_instantiations_the_renderer_launches(solr, probes, fixture):
    """features = 0 picks renderImpl's row for the resident scene: the lean sphere + plane kernel for the Cornell room,
    sphere + cylinder for the sticks, sphere + triangle for the meshes, everything for the mixed and textured scenes"""
    expect = {"closest_cornell": E.F_SPHERE | E.F_PLANE, "closest_sticks": E.F_SPHERE | E.F_CYL,
              "closest_triangles": E.F_SPHERE | E.F_TRI, "closest_mix": E.EVERYTHING, "closest_textured": E.EVERYTHING}
    for name, features in expect.items():
        case, _ = _load(probes, fixture, name)
        assert E.engine_outputs(solr, case)["features"] == features, name


def test_the_probe_entry_points_are_exported(solr):
    """(CPU) include/solr_hip_probes.h is test-only, but what it declares must be in the library"""
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "solr_hip_probes.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(solr_hip_probe_\w+)\s*\(", text)))
    assert len(names) == 9
    hip = solr.hip_lib()
    assert all(hasattr(hip, n) for n in names), [n for n in names if not hasattr(hip, n)]
