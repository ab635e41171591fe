"""k_volumeRenderer (CudaRayTracer.cu:592-713, launchVolumeRendering :50-67, intersectionsWithPrimitives
GeometryIntersections.cuh:1088-1265), the kernel the reference launches for cameraType == ctVolumeRendering
(CudaRayTracer.cu:1777-1806): every primitive along the ray beyond postProcessingInfo.param1 is shaded as a first hit
and kept in ten depth-sorted layers, which are composited front to back with weight 1 / param2.

The HIP engine renders that camera inside the full-featured instantiations of its renderer kernel (rt_device.h
launchVolumeRendering); here it is held to the oracle's restatement (oracle/solr_oracle.c
volumeIntersectionsWithPrimitives / volumeRendererPixel) at the bar of every other camera: ids exact, RGB8 exact,
float colour <= 1 ULP - on scenes of every primitive type, with and without shading, through refinement and
accumulation passes, on strips, and with more than ten primitives along a ray (the layers overflow)."""
import importlib

import numpy as np
import pytest

import scenes_extra as X
from helpers import assert_frame_pinned, assert_parity, assert_pass_parity, compare_frames, gpu_frame, oracle_frame

pytestmark = pytest.mark.gpu
solr_mod = importlib.import_module("sol-r_amd")


def _volume(k, threshold=0.0, density=10.0, **info):
    k.set_scene_info(cameraType=solr_mod.ctVolumeRendering, **info)
    k.set_post_processing(type=solr_mod.ppe_none, param1=threshold, param2=density, param3=0)


def _both(solr, oracle, build, threshold=0.0, density=10.0, info=None, **kw):
    k = solr.Kernel(engine="hip")
    build(k, **kw)
    _volume(k, threshold, density, **(info or {}))
    pp, ids, rgb = gpu_frame(k)
    k.check(0, "volume frame")
    opp, oids, orgb, counts, status = oracle_frame(k, oracle)
    assert status == 0
    res = compare_frames(pp, ids, rgb, opp, oids, orgb)
    res["lit"] = float((pp[..., :3].sum(axis=-1) > 0).mean())
    return k, res, (pp, ids, rgb)


@pytest.mark.parametrize("threshold,density", [(0.0, 10.0), (9000.0, 4.0), (16000.0, 25.0)],
                         ids=["everything", "beyond-9000", "beyond-16000"])
def test_cornell_layers(solr, oracle, threshold, density):
    k, res, frame = _both(solr, oracle, solr.scenes.cornell, threshold, density, width=160, height=120, iterations=2)
    k.finalize()
    assert res["lit"] > 0.5, res
    assert_parity(res)
    # CRT:59-61: the ids are (-1, 1, 0, untouched)
    ids = frame[1]
    assert (ids[..., 0] == -1).all() and (ids[..., 1] == 1).all() and (ids[..., 2] == 0).all()
    # CRT:633, 686-687: the depth written on pass 0 is the kernel's initial 0
    assert (frame[0][..., 3] == 0).all()


def test_without_shading(solr, oracle):
    # GI:1176: glNoShading keeps the material colour
    k, res, _ = _both(solr, oracle, solr.scenes.cornell, info=dict(graphicsLevel=solr_mod.glNoShading), width=128,
                      height=96, iterations=1)
    k.finalize()
    assert res["lit"] > 0.5
    assert_parity(res)


def test_every_primitive_type(solr, oracle):
    # against the oracle as pinned: the procedural sphere's cos / sin of the hit point are glibc's there, and the
    # pixels outside the bar are counted and carry the oracle's mark (test_gpu_parity.py PROCEDURAL_EXCEPTIONS; this
    # camera shades every layer along a ray as a first hit, so more rays meet the sphere)
    k = solr.Kernel(engine="hip")
    X.primitives_mix(k)
    _volume(k, 0.0, 6.0)
    frame = gpu_frame(k)
    k.check(0, "volume frame")
    res = assert_frame_pinned(k, oracle, frame, 16, "volume camera, every primitive type", marked_bounds=(None, 2))
    k.finalize()
    assert float((frame[0][..., :3].sum(axis=-1) > 0).mean()) > 0.2, res


@pytest.mark.parametrize("scene", [X.triangles_only, X.sticks])
def test_meshes_and_sticks(solr, oracle, scene):
    k, res, _ = _both(solr, oracle, scene, density=8.0)
    k.finalize()
    assert res["lit"] > 0.05, res
    assert_parity(res)


def _row_of_spheres(k, count=16, width=96, height=64, **info):
    """more primitives along the central rays than there are layers: the insertion pushes the farthest out"""
    k.initialize(width=width, height=height, nbRayIterations=1, **info)
    rng = solr_mod.scenes.LCG(11)
    for i in range(count):
        m = k.add_material(rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), transparency=0.5,
                           specValue=0.4, specPower=30.0)
        k.add_primitive(solr_mod.ptSphere, (rng.uniform(-300, 300), rng.uniform(-300, 300), -4000.0 + 900.0 * i),
                        size=(1500.0 - 40.0 * i, 0, 0), material=m)
    solr_mod.scenes.add_light(k)
    k.compact_boxes(True)
    k.set_camera((0.0, 0.0, -15000.0))
    return k


def test_more_than_ten_layers(solr, oracle):
    k, res, frame = _both(solr, oracle, _row_of_spheres, density=12.0)
    k.finalize()
    assert res["lit"] > 0.1
    assert_parity(res)


def test_refinement_and_accumulation_passes(solr, oracle):
    # CRT:670-671: the rotated-grid offset of pass % 4 on EVERY pass; CRT:617-629: depth-of-field jitter from
    # pass NB_MAX_ITERATIONS on (with the pass-0 depth: 0); CRT:689-712: store, then max + accumulate
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=96, height=64, iterations=1)
    _volume(k, 2000.0, 10.0)
    opp = oids = None
    previous = None
    frames = []
    for it in range(0, 14):
        k.set_scene_info(pathTracingIteration=it, maxPathTracingIterations=20)
        pp, ids, rgb = gpu_frame(k)
        if previous is not None:    # one pass over the engine's previous buffers: the bar itself
            assert_pass_parity(k, oracle, (pp, ids, rgb), previous, what="volume camera, pass %d" % it)
        previous = (pp.copy(), ids.copy())
        opp, oids, orgb, counts, status = oracle_frame(k, oracle, pp=opp, ids=oids)
        assert status == 0
        res = compare_frames(pp, ids, rgb, opp, oids, orgb)
        assert_parity(res, max_ulp=2)
        frames.append(pp[..., 4:7].copy())
    k.finalize()
    # the offset moves the image from pass to pass (it is not only applied from pass 10 on)
    assert not np.array_equal(frames[0], frames[1])


def test_random_illumination(solr, oracle):
    k, res, _ = _both(solr, oracle, solr.scenes.cornell, info=dict(advancedIllumination=solr_mod.aiRandomIllumination),
                      width=96, height=64, iterations=1)
    k.finalize()
    assert_parity(res)


def test_strips_assemble_to_the_frame(solr, oracle):
    """a process renders rows [first, first + n) of the frame (solr_hip_set_strip): the strips of a volume frame are
    the rows of the whole one"""
    import ctypes as C
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=96, height=72, iterations=1)
    _volume(k, 0.0, 10.0)
    whole_pp, whole_ids, whole_rgb = gpu_frame(k)
    try:
        for first, rows in ((0, 24), (24, 17), (41, 31)):
            hip.solr_hip_set_strip(first, rows)
            k.render()
            k.check(0, "strip")
            spp = np.zeros((rows, 96, 8), np.float32)
            hip.solr_hip_d2h_postprocessing(C.c_void_p(spp.ctypes.data))
            assert np.array_equal(spp.view(np.uint32), whole_pp[first:first + rows].view(np.uint32)), (first, rows)
    finally:
        hip.solr_hip_set_strip(0, -1)
        k.finalize()


def test_other_cameras_are_untouched_by_a_volume_frame(solr, oracle):
    """the volume frame uses eleven colour-stack slots and the exact node list: the next perspective frame of the
    same kernel is the one it would have been"""
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=96, height=64, iterations=3)
    before = gpu_frame(k)
    _volume(k, 0.0, 10.0)
    gpu_frame(k)
    k.set_scene_info(cameraType=solr_mod.ctPerspective)
    k.set_post_processing(type=solr_mod.ppe_none, param1=0.0, param2=0.0, param3=0)
    after = gpu_frame(k)
    k.finalize()
    assert np.array_equal(before[0][..., :3].view(np.uint32), after[0][..., :3].view(np.uint32))
    assert np.array_equal(before[2], after[2])
