"""The C-ABI library: loads, exports everything include/solr_hip.h declares, record sizes match
the reference's (CUDA flavour), and without a GPU it fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:[A-Za-z_][\w \*]*?)\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M)))


def test_every_declared_symbol_is_exported(solr):
    names = declared_functions("solr_hip.h")
    reference_boundary = {"initialize_scene", "finalize_scene", "reshape_scene", "h2d_scene", "h2d_materials",
                          "h2d_randoms", "h2d_textures", "h2d_lightInformation", "d2h_bitmap", "cudaRender"}
    assert reference_boundary <= set(names), "the ten entry points of CudaRayTracer.h:25-67"
    assert len(names) >= 30
    lib = solr.hip_lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_host_library_exports_the_flat_api(solr):
    lib = solr.host_lib()
    for n in ("SolR_SetSceneInfo", "SolR_SetPostProcessingInfo", "SolR_SetDraftMode", "SolR_InitializeKernel",
              "SolR_FinalizeKernel", "SolR_ResetKernel", "SolR_SetCamera", "SolR_RunKernel", "SolR_AddPrimitive",
              "SolR_SetPrimitive", "SolR_GetPrimitive", "SolR_GetPrimitiveAt", "SolR_GetPrimitiveCenter",
              "SolR_RotatePrimitives", "SolR_SetPrimitiveMaterial", "SolR_GetPrimitiveMaterial",
              "SolR_SetPrimitiveNormals", "SolR_SetPrimitiveTextureCoordinates", "SolR_AddMaterial",
              "SolR_SetMaterial", "SolR_CompactBoxes", "SolR_GetLight", "SolR_SetTexture", "SolR_GetTextureSize",
              "SolR_GetNbTextures"):
        assert hasattr(lib, n), n


def test_record_layouts(solr):
    # sizes of the CUDA-flavour records; the C side pins the same with static asserts
    assert C.sizeof(solr.SceneInfo) == 112 and solr.SceneInfo.backgroundColor.offset == 96
    assert solr.SceneInfo.geometryEpsilon.offset == 88 and solr.SceneInfo.pathTracingIteration.offset == 40
    assert C.sizeof(solr.PostProcessingInfo) == 16
    assert solr.BOX_DTYPE.itemsize == 48 and solr.PRIMITIVE_DTYPE.itemsize == 128
    assert solr.MATERIAL_DTYPE.itemsize == 176 and solr.LIGHT_DTYPE.itemsize == 48 and solr.PP_DTYPE.itemsize == 32
    assert solr.PRIMITIVE_DTYPE.fields["type"][1] == 84 and solr.PRIMITIVE_DTYPE.fields["vt0"][1] == 96
    assert solr.MATERIAL_DTYPE.fields["attributes"][1] == 64 and solr.MATERIAL_DTYPE.fields["mappingOffset"][1] == 160


def test_no_gpu_means_a_loud_failure_not_a_fallback(solr, have_gpu):
    if have_gpu:
        pytest.skip("a GPU is present")
    hip = solr.hip_lib()
    hip.solr_hip_clear_error()
    assert hip.solr_hip_device_count() == 0
    si = solr.SceneInfo()
    si.size_x, si.size_y = 8, 8
    hip.solr_hip_initialize(C.byref(si))
    buf = C.create_string_buffer(256)
    assert hip.solr_hip_last_error(buf, 256) != 0 and b"no HIP device" in buf.value
    # every later call is a no-op while the error is pending
    hip.solr_hip_reshape(C.byref(si))
    assert hip.solr_hip_last_error(None, 0) != 0
    hip.solr_hip_clear_error()
    k = solr.Kernel(engine="hip")
    with pytest.raises(solr.SolrError):
        k.initialize(width=8, height=8)
    hip.solr_hip_clear_error()


def test_calls_before_initialisation_are_rejected(solr, have_gpu):
    if have_gpu:
        pytest.skip("state of a live engine is exercised by the gpu tests")
    hip = solr.hip_lib()
    hip.solr_hip_clear_error()
    hip.h2d_randoms(0, np.zeros(4, np.float32).ctypes.data)
    buf = C.create_string_buffer(256)
    assert hip.solr_hip_last_error(buf, 256) == -1 and b"initialize_scene" in buf.value
    hip.solr_hip_clear_error()


def test_the_short_ray_list_switch_is_state_only(solr):
    """solr_hip_set_short_ray_lists (include/solr_hip.h): a switch of the engine, no compute - forced on / off it says so,
    left to the engine it follows the frames in flight (none here: off)"""
    lib = solr.hip_lib()
    try:
        lib.solr_hip_set_short_ray_lists(1)
        assert lib.solr_hip_short_ray_lists() == 1
        lib.solr_hip_set_short_ray_lists(0)
        assert lib.solr_hip_short_ray_lists() == 0
        lib.solr_hip_set_short_ray_lists(-1)
        if not os.environ.get("SOLR_HIP_SHORT_RAY_LISTS"):
            assert lib.solr_hip_short_ray_lists() == 0
    finally:
        lib.solr_hip_set_short_ray_lists(-1)


def test_read_back_tickets_are_positive_for_ever(solr):
    """ a ticket of solr_hip_d2h_image_async is (serial mod period) x 6 + slot: a positive int for
    every 64-bit serial - the old (int)(serial x 6 + slot) went negative after 2^31 / 6 frames, under four hours of an
    eight-rank job - with the slot in its low part, the generation above it, and no two of any 2 x 6 consecutive
    serials alike.  Plain arithmetic: no GPU"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import engine_probes as E
    hip = solr.hip_lib()
    E.declare(hip)
    slot, period = C.c_int(), C.c_longlong()
    limit = (1 << 31) // 6
    for base in (1, 5, limit - 20, limit, (1 << 31) - 7, (1 << 40), (3 << 40) + 12345, (1 << 62) + 99):
        tickets = []
        for serial in range(base, base + 12):
            t = hip.solr_hip_probe_ticket(C.c_longlong(serial), C.byref(slot), C.byref(period))
            assert 0 <= t < (1 << 31), (serial, t)
            assert slot.value == serial % 6 == t % 6
            assert t // 6 == serial % period.value
            tickets.append(t)
        assert len(set(tickets)) == 12
    assert period.value % 6 == 0 and (period.value + 6) * 6 <= (1 << 31) - 1 and period.value > (1 << 31) // 6 // 2 // 3
    # the generation is the process's own serial, never reset (not by solr_hip_image_share either): two tickets of one
    # process are alike only a whole period - 357 million tickets - apart
    a = hip.solr_hip_probe_ticket(C.c_longlong(1234567), None, None)
    assert hip.solr_hip_probe_ticket(C.c_longlong(1234567 + period.value), None, None) == a
    assert all(hip.solr_hip_probe_ticket(C.c_longlong(1234567 + d), None, None) != a for d in (6, 600, 6000006, period.value - 6))
