"""ctypes loader for the CPU oracle (oracle/libsolr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by the sol-r_amd package.
Parity status (pinned bit for bit to the reference's own functions through oracle/ref_probes.cl): solr_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libsolr_oracle.so")


class OracleScene(C.Structure):
    _fields_ = [("boxes", C.c_void_p), ("nbBoxes", C.c_int),
                ("primitives", C.c_void_p), ("nbPrimitives", C.c_int),
                ("lights", C.c_void_p), ("nbLights", C.c_int), ("nbLamps", C.c_int),
                ("materials", C.c_void_p), ("textures", C.c_void_p), ("randoms", C.c_void_p),
                ("nbRandoms", C.c_long)]


def build():
    res = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + res.stdout + res.stderr)
    return LIB


_lib = None
_use_coverage = False
COVERAGE_LIB = os.path.join(_HERE, "libsolr_oracle_cov.so")


def use_coverage_build():
    """tests/test_cuda_text_model.py, before the library is first loaded in its process: load the build with a
    counter on every dialect switch (make -C oracle coverage) instead of the normal one"""
    global _use_coverage
    if _lib is not None and not _use_coverage:
        raise RuntimeError("the oracle is already loaded without the site counters")
    res = subprocess.run(["make", "-C", _HERE, "coverage"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("oracle coverage build failed:\n" + res.stdout + res.stderr)
    _use_coverage = True


def site_hits(reset=False):
    """[(hits in the CUDA dialect, hits in the OpenCL dialect)] per dialect switch of solr_oracle.c, in source
    order; empty with the normal build"""
    L = lib()
    hits = (C.c_ulong * 256)()
    n = L.oracle_site_hits(hits, 256, 1 if reset else 0)
    return [(int(hits[2 * i]), int(hits[2 * i + 1])) for i in range(n)]


def lib():
    global _lib
    if _lib is None:
        if not _use_coverage and not os.path.exists(LIB):
            build()
        L = C.CDLL(COVERAGE_LIB if _use_coverage else LIB)
        L.oracle_render.argtypes = [C.POINTER(OracleScene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int]
        L.oracle_render.restype = C.c_int
        L.oracle_box_intersection.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        L.oracle_box_intersection.restype = C.c_int
        L.oracle_primitive_intersection.argtypes = [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 4
        L.oracle_primitive_intersection.restype = C.c_int
        L.oracle_closest_hit.argtypes = [C.POINTER(OracleScene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_closest_hit.restype = C.c_int
        L.oracle_shadow.argtypes = [C.POINTER(OracleScene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p]
        L.oracle_shadow.restype = C.c_float
        L.oracle_vector_rotation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_make_color.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_set_misround_mask.argtypes = [C.c_void_p, C.c_long]
        L.oracle_set_misround_mask.restype = None
        _lib = L
    return _lib


class Scene:
    """Keeps the numpy arrays alive next to the C view of them."""

    def __init__(self, flat):
        self.keep = [np.ascontiguousarray(a) for a in (flat.boxes, flat.primitives, flat.lights, flat.materials,
                                                        flat.randoms, flat.textures)]
        boxes, prims, lights, mats, rnd, tex = self.keep
        # the reference's material array always has NB_MAX_MATERIALS + 1 records, zeros beyond the active ones,
        # and the box-debug view indexes it with box.startIndex % NB_MAX_MATERIALS (GI:695): give the
        # restatement the same array, not just the active records
        # the texture atlas is followed by zeros, like the engine's (h2d_textures): a material's secondary maps
        # are read at its diffuse texture's texel index, past the end of a smaller map and, for the last
        # textures, past the atlas (undefined in the reference)
        if len(tex):
            tex = self.keep[5] = np.concatenate([tex, np.zeros(len(tex) + 4, np.uint8)])
        capacity = 65506 + 30 + 1
        if len(mats) < capacity:
            full = np.zeros(capacity, dtype=mats.dtype)
            full[: len(mats)] = mats
            mats = self.keep[3] = full
        # numpy silently re-packs padded structured dtypes in some operations (concatenate): insist on the C layout
        assert boxes.dtype.itemsize == 48 and prims.dtype.itemsize == 128 and mats.dtype.itemsize == 176
        assert lights.dtype.itemsize == 48 and rnd.dtype == np.float32 and tex.dtype == np.uint8
        self.c = OracleScene(boxes.ctypes.data, len(boxes), prims.ctypes.data, len(prims), lights.ctypes.data,
                             len(lights), flat.nb_lamps, mats.ctypes.data, tex.ctypes.data if len(tex) else None,
                             rnd.ctypes.data if len(rnd) else None, len(rnd))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def render(flat, scene_info, pp_info, eye, direction, angles, first_row=0, nb_rows=None, pp=None, ids=None,
           nthreads=0, misround=None):
    """oracle_render for rows [first_row, first_row + nb_rows).

    scene_info / pp_info are ctypes structures laid out like include/solr_types.h.
    Returns (pp (rows, W, 8) float32, ids (rows, W, 4) int32, bitmap (rows, W, 3) uint8, counts[4], status).
    misround: a zeroed (rows, W) uint8 array that receives, per pixel, bit 0 where a powf and bit 1 where a
    sinf / cosf / atan2f / asinf of that pixel was not the correctly rounded value (oracle_set_misround_mask).
    """
    L = lib()
    s = Scene(flat)
    w = scene_info.size_x
    rows = scene_info.size_y if nb_rows is None else nb_rows
    pp = np.zeros((rows, w, 8), np.float32) if pp is None else np.ascontiguousarray(pp, np.float32).copy()
    ids = np.zeros((rows, w, 4), np.int32) if ids is None else np.ascontiguousarray(ids, np.int32).copy()
    bitmap = np.zeros((rows, w, 3), np.uint8)
    counts = (C.c_ulonglong * 4)()
    eye, direction, angles = _f(eye), _f(direction), _f(angles)
    if misround is not None:
        assert misround.dtype == np.uint8 and misround.shape == (rows, w) and misround.flags.c_contiguous
        L.oracle_set_misround_mask(C.c_void_p(misround.ctypes.data), C.c_long(rows * w))
    try:
        status = L.oracle_render(C.byref(s.c), C.addressof(scene_info), C.addressof(pp_info), eye.ctypes.data,
                                 direction.ctypes.data, angles.ctypes.data, first_row, rows, pp.ctypes.data,
                                 ids.ctypes.data, bitmap.ctypes.data, C.addressof(counts), nthreads)
    finally:
        if misround is not None:
            L.oracle_set_misround_mask(None, C.c_long(0))
    return pp, ids, bitmap, [int(c) for c in counts], status


def max_threads():
    return lib().oracle_max_threads()


# ---- oracle/_ref: the reference's OpenCL renderer (built by `make -C oracle ref` where /root/reference exists)
REF_DIR = os.path.join(_HERE, "_ref")
REF_LIB = os.path.join(REF_DIR, "libsolr_ref_opencl.so")
REF_CODE_OBJECT = os.path.join(REF_DIR, "RayTracer_gfx950.co")


def build_ref(reference="/root/reference"):
    """Compiles the reference's RayTracer.cl for gfx950 and the OpenCL host runner into oracle/_ref/.
    Returns False (and builds nothing) where the reference tree is absent, e.g. on the GPU box."""
    if not os.path.exists(os.path.join(reference, "solr/engines/opencl/RayTracer.cl")):
        return False
    res = subprocess.run(["make", "-C", _HERE, "ref", "REFERENCE=" + reference], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("oracle/_ref build failed:\n" + res.stdout + res.stderr)
    return True


def have_ref():
    return os.path.exists(REF_LIB) and os.path.exists(REF_CODE_OBJECT)


def ref_render(flat, scene_info, pp_info, eye, direction, angles, repeats=1, timing=None):
    """One frame of the reference's OpenCL k_standardRenderer + k_default on the GPU (ref_opencl_runner.c).
    Returns (pp (H, W, 8) float32, ids (H, W, 4) int32, bitmap (H, W, 3) uint8).  repeats > 1 launches the
    renderer that many times and stores the mean kernel milliseconds (OpenCL events, first launch
    excluded) in timing["renderer_ms"]."""
    R = C.CDLL(REF_LIB)
    R.solr_ref_opencl_render.restype = C.c_int
    s = Scene(flat)
    boxes, prims, lights, mats, rnd, _ = s.keep
    w, h = scene_info.size_x, scene_info.size_y
    pp = np.zeros((h, w, 8), np.float32)
    ids = np.zeros((h, w, 4), np.int32)
    rgb = np.zeros((h, w, 3), np.uint8)
    if len(rnd) == 0:
        rnd = np.zeros(16, np.float32)
    eye, direction, angles = _f(eye), _f(direction), _f(angles)
    log = C.create_string_buffer(4096)
    ms = C.c_double(0.0)
    vp = C.c_void_p
    status = R.solr_ref_opencl_render(
        REF_CODE_OBJECT.encode(), vp(boxes.ctypes.data), C.c_int(len(boxes)), vp(prims.ctypes.data),
        C.c_int(len(prims)), vp(lights.ctypes.data), C.c_int(len(lights)), C.c_int(flat.nb_lamps),
        vp(mats.ctypes.data), C.c_int(len(mats)), vp(rnd.ctypes.data), C.c_int(len(rnd)),
        vp(C.addressof(scene_info)), vp(C.addressof(pp_info)), vp(eye.ctypes.data), vp(direction.ctypes.data),
        vp(angles.ctypes.data), vp(pp.ctypes.data), vp(ids.ctypes.data), vp(rgb.ctypes.data), C.c_int(repeats),
        C.byref(ms), log, C.c_int(4096))
    if timing is not None:
        timing["renderer_ms"] = ms.value
        timing["log"] = log.value.decode(errors="replace")
    if log.value and status == 0 and os.environ.get("SOLR_REF_VERBOSE"):
        print("reference runner:", log.value.decode(errors="replace"))
    if status != 0:
        raise RuntimeError("reference OpenCL renderer failed (%d): %s" % (status, log.value.decode(errors="replace")))
    return pp, ids, rgb
