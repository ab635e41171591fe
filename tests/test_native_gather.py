"""The multi-GPU path of the C ABI (include/solr_hip.h: solr_hip_strip_rows, solr_hip_comm_*,
solr_hip_gather_strips): row strips gathered with RCCL called from the library itself, no torch.

CPU: the strip arithmetic equals the launcher's (sol-r_amd.strip_rows).  GPU: a communicator of ONE rank on the
box's one GPU - RCCL's send-to-self inside a group - carries every frame of the loop: full frame, a strip,
several frames in flight; the gathered image must be the rendered one.  (N > 1 needs as many GPUs as ranks; the
world-size-2 logic of the partition and the assembly is covered on CPU by tests/test_strips_gloo.py.)"""
import ctypes as C

import numpy as np
import pytest


def test_strip_rows_of_the_library_are_the_launchers(solr):
    hip = solr.hip_lib()
    first, count, per = C.c_int(), C.c_int(), C.c_int()
    for height in (1, 7, 17, 45, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for rank in range(world):
                hip.solr_hip_strip_rows(rank, world, height, C.byref(first), C.byref(count), C.byref(per))
                assert (first.value, count.value, per.value) == solr.strip_rows(rank, world, height)
                covered += list(range(first.value, first.value + count.value))
            assert covered == list(range(height))


def test_gather_without_a_communicator_is_refused(solr, have_gpu):
    hip = solr.hip_lib()
    hip.solr_hip_clear_error()
    assert hip.solr_hip_gather_strips(0) == -1
    assert hip.solr_hip_last_error(None, 0) != 0
    hip.solr_hip_clear_error()


@pytest.mark.gpu
def test_native_rccl_gather_with_one_rank(solr):
    W, H = 192, 136
    hip = solr.hip_lib()
    hip.solr_hip_gathered_frame.restype = C.c_void_p
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=W, height=H, iterations=2)
    full = k.render()
    k.check(0, "first frame")
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    si.pathTracingIteration = 0
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731

    def render(dx=0.0):
        e = eye.copy()
        e[0] += dx
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(e), fp(direction), fp(angles))

    uid = C.create_string_buffer(128)
    try:
        assert hip.solr_hip_comm_unique_id(uid) == 0
        assert hip.solr_hip_comm_init(0, 1, uid) == 0
        k.check(0, "communicator")
        # the whole frame
        render()
        assert hip.solr_hip_gather_strips(0) == 0
        image = np.zeros((H, W, 3), np.uint8)
        assert hip.solr_hip_d2h_gathered(C.c_void_p(image.ctypes.data)) == 0
        assert np.array_equal(image, full)
        # a strip: lands at its rows of the assembled frame, the other rows keep the frame before
        image[:] = 0
        hip.solr_hip_set_strip(40, 32)
        render(dx=700.0)
        moved = np.zeros((H, W, 3), np.uint8)
        hip.solr_hip_d2h(C.byref(si), C.c_void_p(moved.ctypes.data), None)     # places the strip at its rows
        assert hip.solr_hip_gather_strips(0) == 0
        assert hip.solr_hip_d2h_gathered(C.c_void_p(image.ctypes.data)) == 0
        assert np.array_equal(image[40:72], moved[40:72]) and not np.array_equal(moved[40:72], full[40:72])
        assert np.array_equal(image[:40], full[:40]) and np.array_equal(image[72:], full[72:])
        hip.solr_hip_set_strip(0, -1)
        # three frames in flight, a gather behind every frame on its own stream, no host wait in between
        hip.solr_hip_set_frames_in_flight(3)
        shifts = [100.0 * i for i in range(9)]
        for dx in shifts:
            render(dx)
            assert hip.solr_hip_gather_strips(0) == 0
        assert hip.solr_hip_d2h_gathered(C.c_void_p(image.ctypes.data)) == 0
        hip.solr_hip_set_frames_in_flight(1)
        render(shifts[-1])
        expected = np.zeros((H, W, 3), np.uint8)
        hip.solr_hip_d2h(C.byref(si), C.c_void_p(expected.ctypes.data), None)
        assert np.array_equal(image, expected)
        k.check(0, "gather loop")
    finally:
        hip.solr_hip_set_strip(0, -1)
        hip.solr_hip_set_frames_in_flight(1)
        hip.solr_hip_comm_finalize()
        hip.solr_hip_clear_error()
        k.finalize()
