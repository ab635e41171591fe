"""Shared helpers of the parity tests."""
import ctypes as C

import numpy as np


def ulp_distance(a, b):
    """Element-wise distance in units in the last place between two float32 arrays
    (sign-magnitude aware; a NaN only equals a NaN: where the reference's arithmetic ends in 0/0 - a
    normal map that flattens the normal to zero length, TextureMapping.cuh:30-40 - both sides must)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    d[np.isnan(a) | np.isnan(b)] = 1 << 40
    d[np.isnan(a) & np.isnan(b)] = 0
    return d


def compare_frames(gpu_pp, gpu_ids, gpu_rgb, ora_pp, ora_ids, ora_rgb):
    """Parity figures between an engine frame and an oracle frame."""
    ulp = ulp_distance(gpu_pp[..., :3], ora_pp[..., :3])
    last = ulp_distance(gpu_pp[..., 4:7], ora_pp[..., 4:7])
    depth = ulp_distance(gpu_pp[..., 3], ora_pp[..., 3])
    return {
        "max_ulp": int(ulp.max()) if ulp.size else 0,
        "last_sample_max_ulp": int(last.max()) if last.size else 0,
        "pixels_over_1ulp": int((ulp.max(axis=-1) > 1).sum()) if ulp.size else 0,
        "pixels_nonzero_ulp": int((ulp.max(axis=-1) > 0).sum()) if ulp.size else 0,
        "depth_max_ulp": int(depth.max()) if depth.size else 0,
        "ids_equal": bool(np.array_equal(gpu_ids[..., :2], ora_ids[..., :2])),
        "ids_all_equal": bool(np.array_equal(gpu_ids, ora_ids)),
        "rgb_max_diff": int(np.abs(gpu_rgb.astype(int) - ora_rgb.astype(int)).max()) if gpu_rgb.size else 0,
        "rgb_equal": bool(np.array_equal(gpu_rgb, ora_rgb)),
    }


def assert_parity(res, max_ulp=1, rgb_max_diff=0, depth_ulp=0):
    """The bar of BASELINE.json: ids exact, RGB8 exact, float colour <= 1 ULP.
    rgb_max_diff=1 is allowed only where a 1-ULP float difference straddles a x255 truncation."""
    assert res["ids_all_equal"], res
    assert res["max_ulp"] <= max_ulp, res
    assert res["last_sample_max_ulp"] <= max_ulp, res
    assert res["depth_max_ulp"] <= depth_ulp, res
    assert res["rgb_max_diff"] <= rgb_max_diff, res


def assert_parity_pinned(gpu, ora, misround, max_exceptions, what="", marked_bounds=(2, 1)):
    """The bar against the oracle AS PINNED (libm's binary32 powf / sinf / cosf / atan2f / asinf, not switched to
    the engine's correctly rounded forms): primitive ids, depth exact; float colour <= 1 ULP and RGB8 exact on
    every pixel except a COUNTED set, each member of which (a) is at most 2 ULP / one RGB8 step off and (b) went,
    in the oracle, through a libm result that is not the correctly rounded value - `misround`, filled by
    oracle.render(misround=...), which evaluates the same call in binary64 at the call and rounds once, which is
    what the engine does.  Returns the figures with the number of such pixels.
    marked_bounds = (ULPs, RGB8 steps) a marked pixel may be off; None for a figure that has no bound: a powf result
    one ULP off stays a colour one or two ULP off (the default), but a mis-rounded sinf / cosf moves a procedural
    sphere's hit point, and a mis-rounded atan2f / asinf that lands on the other side of a texel boundary selects the
    NEIGHBOURING texel of every map of the material (diffuse, normal, bump ...: TextureMapping.cuh:30-116 fetches them
    all at the diffuse map's index) - what such a pixel shows is another texel's shading, not a rounding of this one's.
    Those tests say so, pass None and keep the count."""
    gpu_pp, gpu_ids, gpu_rgb = gpu
    ora_pp, ora_ids, ora_rgb = ora
    res = compare_frames(gpu_pp, gpu_ids, gpu_rgb, ora_pp, ora_ids, ora_rgb)
    res["what"] = what
    assert res["ids_all_equal"], res
    assert res["depth_max_ulp"] == 0, res
    ulp = np.maximum(ulp_distance(gpu_pp[..., :3], ora_pp[..., :3]),
                     ulp_distance(gpu_pp[..., 4:7], ora_pp[..., 4:7])).max(axis=-1)
    rgb = np.abs(gpu_rgb.astype(int) - ora_rgb.astype(int)).max(axis=-1)
    outside = (ulp > 1) | (rgb > 0)
    res["pixels_outside_the_bar"] = int(outside.sum())
    res["pixels_with_a_misrounded_libm_result"] = int((misround != 0).sum())
    unexplained = outside & (misround == 0)
    assert not unexplained.any(), (res, "pixels outside the bar that met no mis-rounded libm result",
                                   np.argwhere(unexplained)[:8].tolist())
    if marked_bounds[0] is not None:
        assert (ulp[outside] <= marked_bounds[0]).all(), res
    if marked_bounds[1] is not None:
        assert (rgb[outside] <= marked_bounds[1]).all(), res
    assert res["pixels_outside_the_bar"] <= max_exceptions, res
    return res


def assert_frame_pinned(k, oracle, gpu, max_exceptions, what="", marked_bounds=(2, 1)):
    """the engine's frame `gpu` = (pp, ids, rgb) of kernel k against the oracle as pinned (assert_parity_pinned)"""
    misround = np.zeros(gpu[0].shape[:2], np.uint8)
    assert not oracle.lib().oracle_get_rounded_transcendentals()
    opp, oids, orgb, _, status = oracle_frame(k, oracle, misround=misround)
    assert status == 0, "the oracle read outside the random buffer"
    return assert_parity_pinned(gpu, (opp, oids, orgb), misround, max_exceptions, what, marked_bounds)


def gpu_frame(k):
    rgb = k.render()
    return k.postprocessing_buffer(), k.primitive_ids(), rgb


def oracle_frame(k, oracle, pp=None, ids=None, first_row=0, nb_rows=None, nthreads=0, misround=None):
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    opp, oids, orgb, counts, status = oracle.render(flat, si, ppi, eye, direction, angles, first_row=first_row,
                                                    nb_rows=nb_rows, pp=pp, ids=ids, nthreads=nthreads,
                                                    misround=misround)
    return opp, oids, orgb, counts, status


def assert_pass_parity(k, oracle, gpu, previous, max_exceptions=2, what="", first_row=0, nb_rows=None):
    """ONE pass held to the bar: the oracle (as pinned) renders the pass over the ENGINE's previous buffers
    (`previous` = (pp, ids) of the frame before, None for a first pass), so the running sum of an accumulation takes
    one rounding on either side; ids, depth exact, RGB8 exact, float colour and last sample <= 1 ULP but for the
    counted pixels behind a mis-rounded libm result (assert_parity_pinned)."""
    rows = gpu[0].shape[0]
    misround = np.zeros((rows, gpu[0].shape[1]), np.uint8)
    pp, ids = (None, None) if previous is None else previous
    opp, oids, orgb, _, status = oracle_frame(k, oracle, pp=pp, ids=ids, first_row=first_row, nb_rows=nb_rows,
                                              misround=misround)
    assert status == 0, "the oracle read outside the random buffer"
    return assert_parity_pinned(gpu, (opp, oids, orgb), misround, max_exceptions, what)


def f3(*v):
    return (C.c_float * 3)(*v)


def device_frame(solr, si):
    """Float framebuffer, primitive ids and RGB8 of the frame the engine rendered last, straight from
    the C-ABI (for tests that render through solr_hip_render instead of the host protocol)."""
    hip = solr.hip_lib()
    w, h = si.size_x, si.size_y
    pp = np.zeros((h, w, 8), np.float32)
    ids = np.zeros((h, w, 4), np.int32)
    rgb = np.zeros((h, w, 3), np.uint8)
    hip.solr_hip_d2h_postprocessing(C.c_void_p(pp.ctypes.data))
    hip.solr_hip_d2h(C.byref(si), C.c_void_p(rgb.ctypes.data), C.c_void_p(ids.ctypes.data))
    return pp, ids, rgb
