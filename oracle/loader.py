"""ctypes loader for the CPU oracle (oracle/libsolr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by the sol-r_amd package.
PARITY UNPINNED - see solr_oracle.h.
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


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
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
        _lib = L
    return _lib


class Scene:
    """Keeps the numpy arrays alive next to the C view of them."""

    def __init__(self, flat):
        self.keep = [np.ascontiguousarray(a) for a in (flat.boxes, flat.primitives, flat.lights, flat.materials,
                                                        flat.randoms, flat.textures)]
        boxes, prims, lights, mats, rnd, tex = self.keep
        # numpy silently re-packs padded structured dtypes in some operations (concatenate): insist on the C layout
        assert boxes.dtype.itemsize == 48 and prims.dtype.itemsize == 128 and mats.dtype.itemsize == 176
        assert lights.dtype.itemsize == 48 and rnd.dtype == np.float32 and tex.dtype == np.uint8
        self.c = OracleScene(boxes.ctypes.data, len(boxes), prims.ctypes.data, len(prims), lights.ctypes.data,
                             len(lights), flat.nb_lamps, mats.ctypes.data, tex.ctypes.data if len(tex) else None,
                             rnd.ctypes.data if len(rnd) else None, len(rnd))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def render(flat, scene_info, pp_info, eye, direction, angles, first_row=0, nb_rows=None, pp=None, ids=None,
           nthreads=0):
    """oracle_render for rows [first_row, first_row + nb_rows).

    scene_info / pp_info are ctypes structures laid out like include/solr_types.h.
    Returns (pp (rows, W, 8) float32, ids (rows, W, 4) int32, bitmap (rows, W, 3) uint8, counts[4], status).
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
    status = L.oracle_render(C.byref(s.c), C.addressof(scene_info), C.addressof(pp_info), eye.ctypes.data,
                             direction.ctypes.data, angles.ctypes.data, first_row, rows, pp.ctypes.data,
                             ids.ctypes.data, bitmap.ctypes.data, C.addressof(counts), nthreads)
    return pp, ids, bitmap, [int(c) for c in counts], status


def max_threads():
    return lib().oracle_max_threads()
