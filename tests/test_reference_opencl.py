"""Pins the oracle (and the HIP engine) against OUTPUTS OF THE REFERENCE ITSELF.

oracle/_ref holds the reference's own per-pixel renderer - k_standardRenderer + k_default of its
OpenCL engine, compiled for gfx950 from solr/engines/opencl/RayTracer.cl where it lies under
/root/reference (oracle/Makefile, target `ref`; a binary, no source travels) - and these tests run it
on the MI355X next to the oracle and the engine on the same scenes.

The OpenCL engine is an older sibling of the CUDA engine that the oracle restates, so the comparison
is at image level, with the tolerances written below, not bit for bit.  Known drift, avoided or
compensated here: it jitters the primary ray on every pass (RayTracer.cl:2526-2527; undone by moving
the look-at point, exact only for an unrotated camera), it computes in float4 with fused dot products
(1-ULP noise everywhere, and different outcomes where a hit sits on an epsilon: silhouettes), it
handles a zero direction component differently (off-axis camera), its transparent-shadow term is
weaker by 1-3 % (scenes without glass), and its plane test leaves the .w lanes of the float4 normal
and hit point unwritten (RayTracer.cl:1151-1290 assigns .x .y .z only) while its float4 dot products
read them: the shading of planes depends on what the register held before and changes from run to
run, in whole 8x8 work-groups at a time, and through the float4 length of the hit distance even
which primitive wins (a clean device gives 99.998 % equal ids and 98.9 % identical RGB8 on the full
Cornell room, a device other kernels have run on 94-98 % and 74-95 %).  The pin is therefore taken
on plane-free scenes.  What the
figures below pin: visibility (box walk + every intersection routine, planes included), first-hit
depth, and the shaded colour through diffuse/specular/shadow/reflection passes.
"""
import importlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenes_extra  # noqa: E402

pytestmark = pytest.mark.gpu


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _build(solr, spec, engine):
    name, kw = spec
    k = solr.Kernel(engine=engine, deterministic_seed=1)
    fn = getattr(solr.scenes, name, None) or getattr(scenes_extra, name)
    if "file" in kw:        # a scene loaded from one of the reference's sample files (tests/test_scene_files.py)
        kw = dict(kw)
        fn(k, os.path.join(GOLDEN, kw.pop("file")), **kw)
    else:
        fn(k, **kw)
    # off-axis: no pixel gets an exactly zero direction component
    k.set_camera((131.0, 77.0, -15000.0), look_at=(57.0, 23.0, 0.0))
    return k


def _reference_frame(oracle, k):
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    d = np.array(direction, np.float32).copy()
    d[0] -= np.float32(3.0)   # RayTracer.cl:2474,2526-2527: AArotatedGrid[0] = (3, 5) is added to the
    d[1] -= np.float32(5.0)   # rotated direction on pass 0; the camera here is unrotated
    return oracle.ref_render(flat, si, ppi, eye, d, angles)


def _agreement(pp, ids, rgb, rpp, rids, rrgb):
    same = float((ids[..., 0] == rids[..., 0]).mean())
    diff = np.abs(rgb.astype(int) - rrgb.astype(int)).max(axis=2)
    colour = np.abs(pp[..., :3] - rpp[..., :3]).max(axis=2) / np.maximum(np.abs(pp[..., :3]).max(axis=2), 1e-3)
    depth = np.abs(pp[..., 3] - rpp[..., 3]) / np.maximum(np.abs(pp[..., 3]), 1.0)
    return {"ids_equal": same, "rgb_identical": float((diff == 0).mean()), "rgb_within_8": float((diff <= 8).mean()),
            "colour_within_1e-5": float((colour <= 1e-5).mean()), "colour_median_rel": float(np.median(colour)),
            "depth_median_rel": float(np.median(depth))}


def _near_discontinuity(ids, depth, rgb, reach=1):
    """pixels within `reach` of a place where the primitive under the pixel changes, the first-hit depth jumps
    by more than 1 % or the colour by more than 8 levels: silhouettes, creases and shadow edges, where a hit
    (of the view ray or of the shadow ray) that sits on an epsilon can go either way"""
    prim = ids[..., 0]
    level = rgb.astype(int)
    edge = np.zeros(prim.shape, bool)
    for axis in (0, 1):
        a = np.take(prim, range(1, prim.shape[axis]), axis=axis)
        b = np.take(prim, range(0, prim.shape[axis] - 1), axis=axis)
        da = np.take(depth, range(1, prim.shape[axis]), axis=axis)
        db = np.take(depth, range(0, prim.shape[axis] - 1), axis=axis)
        ca = np.take(level, range(1, prim.shape[axis]), axis=axis)
        cb = np.take(level, range(0, prim.shape[axis] - 1), axis=axis)
        step = (a != b) | (np.abs(da - db) > 0.01 * np.maximum(np.abs(da), np.abs(db))) | \
               (np.abs(ca - cb).max(axis=2) > 8)
        pad = [(0, 0), (0, 0)]
        pad[axis] = (0, 1)
        edge |= np.pad(step, pad)
        pad[axis] = (1, 0)
        edge |= np.pad(step, pad)
    out = edge.copy()
    for _ in range(reach):
        grown = out.copy()
        grown[1:] |= out[:-1]
        grown[:-1] |= out[1:]
        grown[:, 1:] |= out[:, :-1]
        grown[:, :-1] |= out[:, 1:]
        out = grown
    return out


def _unexplained(pp, ids, rgb, rpp, rids, rrgb):
    """(pixels under a different primitive that are NOT at a discontinuity of the reference frame,
        pixels of a clearly different colour that are neither at a discontinuity of either frame nor show a
        reflection or refraction - more than one bounce - which carries such a pixel's difference along)"""
    edge = _near_discontinuity(rids, rpp[..., 3], rrgb) | _near_discontinuity(ids, pp[..., 3], rgb)
    other_primitive = (ids[..., 0] != rids[..., 0]) & ~edge
    diff = np.abs(rgb.astype(int) - rrgb.astype(int)).max(axis=2) > 8
    bounced = (ids[..., 1] > 1) | (rids[..., 1] > 1)
    return int(other_primitive.sum()), int((diff & ~edge & ~bounced).sum())


# scene, minimum fraction of pixels: same primitive, identical RGB8, RGB8 within 8 levels, float colour within 1e-5
CASES = [
    (("cornell", dict(width=256, height=192, iterations=1, glass=0, room=False)), 0.9995, 0.985, 0.99, 0.96),
    (("cornell", dict(width=256, height=192, iterations=3, glass=0, room=False)), 0.9995, 0.975, 0.99, 0.95),
    (("height_field", dict(n=24, width=128, height=96)), 0.9995, 0.985, 0.99, 0.985),
    (("triangles_only", dict(width=80, height=64)), 0.999, 0.999, 0.999, 0.999),
    (("sticks", dict(width=80, height=64)), 0.99, 0.96, 0.965, 0.85),
    # the reference's own sample files, loaded by the readers of sol-r_amd/host (tests/test_scene_files.py)
    (("obj_model", dict(file="cornell.obj", width=256, height=192, iterations=3)), 0.9995, 0.995, 0.995, 0.65),
    (("irt_model", dict(file="model_subset.irt", width=256, height=192, iterations=2, floor=False)), 0.9995, 0.98, 0.985, 0.97),
    (("pdb_molecule", dict(file="1BNA.pdb", width=256, height=192, iterations=2, geometry_type=3)), 0.998, 0.985, 0.988, 0.90),
    (("swc_morphology", dict(file="pyramidal.swc", width=256, height=192, iterations=2)), 0.997, 0.985, 0.988, 0.96),
]


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref is not built (needs /root/reference: python -c 'import __graft_entry__ as g; g.build()')")
    return oracle


@pytest.mark.parametrize("case", CASES, ids=[c[0][0] + "-" + "-".join("%s%s" % kv for kv in c[0][1].items() if kv[0] != "file") for c in CASES])
def test_oracle_reproduces_the_reference_renderer(solr, ref, case):
    spec, min_ids, min_rgb, min_rgb8, min_colour = case
    k = _build(solr, spec, "host-only")
    rpp, rids, rrgb = _reference_frame(ref, k)
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    opp, oids, orgb, _, status = ref.render(flat, si, ppi, eye, direction, angles, nthreads=8)
    assert status == 0
    # the same frame in the oracle's OpenCL dialect (solr_oracle.c): the statements in which the two engines
    # differ switched to the OpenCL engine's form, so that what is left is arithmetic, not drift
    L = ref.lib()
    L.oracle_set_dialect(1)
    try:
        cpp, cids, crgb, _, status = ref.render(flat, si, ppi, eye, direction, angles, nthreads=8)
    finally:
        L.oracle_set_dialect(0)
    assert status == 0
    # what then still differs is explained pixel by pixel: it sits on a silhouette, crease or shadow edge of
    # one of the two frames (the reference renderer as built fuses its dot products: a hit on an epsilon goes
    # either way), or shows one through a reflection.  tests/test_reference_probes.py has the bit-for-bit form.
    dialect = _agreement(cpp, cids, crgb, rpp, rids, rrgb)
    assert _unexplained(cpp, cids, crgb, rpp, rids, rrgb) == (0, 0), dialect
    assert dialect["ids_equal"] >= min_ids and dialect["rgb_within_8"] >= min_rgb8, dialect
    res = _agreement(opp, oids, orgb, rpp, rids, rrgb)
    assert res["ids_equal"] >= min_ids, res
    assert res["rgb_identical"] >= min_rgb, res
    assert res["rgb_within_8"] >= min_rgb8, res
    assert res["colour_within_1e-5"] >= min_colour, res
    assert res["colour_median_rel"] <= 1e-5, res
    # ... and WHY the rest differs: a pixel under another primitive sits on a silhouette of the reference's
    # frame; a pixel of another colour sits on one, or shows one through a reflection
    # first-hit depth: the OpenCL engine measures it from the LAST ray origin of the path and for every
    # pixel (RayTracer.cl:2411-2417), the CUDA engine from the eye and only where something was hit
    # (CudaRayTracer.cu:107,155): comparable on single-bounce frames only
    if spec[1].get("iterations") == 1:
        assert res["depth_median_rel"] <= 1e-6, res


@pytest.mark.parametrize("spec,min_ids,min_rgb", [
    (("cornell", dict(width=256, height=192, iterations=3, glass=0, room=False)), 0.9995, 0.975),
    (("obj_model", dict(file="cornell.obj", width=256, height=192, iterations=3)), 0.9995, 0.995),
    (("pdb_molecule", dict(file="1BNA.pdb", width=256, height=192, iterations=2, geometry_type=3)), 0.998, 0.985),
], ids=["cornell-spheres", "cornell.obj", "1BNA.pdb"])
def test_engine_reproduces_the_reference_renderer(solr, ref, have_gpu, spec, min_ids, min_rgb):
    """The shipped HIP path against the reference renderer directly, no oracle in between."""
    assert have_gpu
    from helpers import gpu_frame
    k = _build(solr, spec, "hip")
    rpp, rids, rrgb = _reference_frame(ref, k)   # before the engine advances its pass counter
    gpp, gids, grgb = gpu_frame(k)
    k.check(0, "render")
    res = _agreement(gpp, gids, grgb, rpp, rids, rrgb)
    assert res["ids_equal"] >= min_ids and res["rgb_identical"] >= min_rgb and res["colour_median_rel"] <= 1e-5, res
    k.finalize()
