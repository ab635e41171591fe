"""The walk's own ceiling (include/solr_hip.h solr_hip_walk_bound; SURVEY.md 8d's "measured empty-traversal upper
bound"): a frame that records its walks is a frame like any other - same bits - and the replay of those walks with the
node loop alone visits leaves, leaves nothing out, and takes less time than the frame."""
import ctypes as C

import numpy as np
import pytest

from helpers import device_frame

pytestmark = pytest.mark.gpu


def _args(solr, k):
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    si.pathTracingIteration = 0
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    return si, objects, ppi, eye, direction, angles


def _walk_bound(solr, args, repeats=5):
    si, objects, ppi, eye, direction, angles = args
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    ms, stats = (C.c_double * 3)(), (C.c_ulonglong * 4)()
    status = solr.hip_lib().solr_hip_walk_bound(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles),
                                                repeats, ms, stats)
    return status, list(ms), [int(x) for x in stats]


@pytest.mark.parametrize("scene", ["cornell", "height_field", "molecule"])
def test_a_recorded_frame_is_the_frame_and_its_walks_replay(solr, scene):
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    W, H = 320, 200
    if scene == "cornell":
        solr.scenes.cornell(k, width=W, height=H, iterations=3)
    elif scene == "height_field":
        solr.scenes.height_field(k, n=48, width=W, height=H)
    else:
        solr.scenes.molecule(k, atoms=1500, width=W, height=H)
    try:
        for _ in range(3):
            k.render()
        args = _args(solr, k)
        si, objects, ppi, eye, direction, angles = args
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        plain = device_frame(solr, si)
        status, ms, stats = _walk_bound(solr, args)
        k.check(status, "solr_hip_walk_bound")
        recorded = device_frame(solr, si)          # the frame solr_hip_walk_bound rendered while recording
        assert np.array_equal(recorded[0].view(np.uint32), plain[0].view(np.uint32))
        assert np.array_equal(recorded[1], plain[1]) and np.array_equal(recorded[2], plain[2])
        walks, left_out, entries, workgroups = stats
        tiles = ((W + 7) // 8) * ((H + 7) // 8)
        assert workgroups >= tiles
        assert walks >= 2 * tiles * 0.5          # at least a closest-hit and a shadow walk for the tiles that see the scene
        assert left_out == 0, "walks the replay left out (more than %d per wave, or not through the node loop)" % 16
        assert entries > W * H * 0.5             # the replay entered leaves
        assert 0.0 < ms[2] <= ms[1]
        # the frame after: the engine is back to plain frames
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        again = device_frame(solr, si)
        assert np.array_equal(again[0].view(np.uint32), plain[0].view(np.uint32))
        k.check(0, "after the replay")
    finally:
        k.finalize()


def test_scenes_outside_the_lean_kernels_are_refused(solr):
    import scenes_extra as X
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    X.textured(k, width=96, height=64)
    try:
        k.render()
        status, _, _ = _walk_bound(solr, _args(solr, k), repeats=1)
        assert status == -1
        buf = C.create_string_buffer(512)
        assert hip.solr_hip_last_error(buf, 512) == -1 and b"no recording instantiation" in buf.value
        hip.solr_hip_clear_error()
        assert k.render().any()
    finally:
        hip.solr_hip_clear_error()
        k.finalize()
