"""The engine's own device functions over the arrays of the reference probes (test infrastructure).

`engine_outputs(solr, case)` uploads what a case of oracle/probes.py needs through the C ABI (h2d_materials,
h2d_textures, h2d_scene, h2d_lightInformation - no host builder, no renderer) and evaluates the engine's device
functions on the case's arrays through the test-only entry points of include/solr_hip_probes.h
(sol-r_amd/csrc/solr_probes.hip).  The result is laid out like oracle.probes.reference_outputs /
oracle_outputs, so that the three can be compared key by key: tests/test_engine_probes_gpu.py holds the engine to
THE REFERENCE'S outputs (tests/golden/reference_probes.npz, source-order build) with no oracle in between.
"""
import ctypes as C

import numpy as np

f32, i32 = np.float32, np.int32

# enum Feature of sol-r_amd/csrc/rt_device.h
F_SPHERE, F_PROC, F_CYL, F_ELL, F_TRI, F_PLANE, F_TEX, F_FULL, F_DEEP = 1, 2, 4, 8, 16, 32, 64, 128, 256
EVERYTHING = (255 & ~F_FULL) | F_DEEP
NOBODY = -12345          # a Primitive.index no primitive has
MAX_BITMAP_SIZE = 1920 * 1080   # Consts.h:39-41


class TextureInfo(C.Structure):
    _fields_ = [("buffer", C.c_void_p), ("offset", C.c_int), ("size", C.c_int * 3), ("type", C.c_int), ("pad", C.c_int)]


assert C.sizeof(TextureInfo) == 32


def _p(a):
    return C.c_void_p(a.ctypes.data)


def declare(hip):
    P = C.POINTER
    v, i = C.c_void_p, C.c_int
    hip.solr_hip_probe_box.argtypes = [i, v, v, v, v, v, v, v]
    hip.solr_hip_probe_box_walk.argtypes = [v, i, v, v, v, i, v]
    hip.solr_hip_probe_primitive.argtypes = [v, i, v, v, v, i, v, v, v, v, v]
    hip.solr_hip_probe_closest.argtypes = [v, i, v, v, v, v, i, i, v, v, v, v, v]
    hip.solr_hip_probe_shadow.argtypes = [v, i, v, v, v, v, v, i, i, v, v]
    hip.solr_hip_probe_shader.argtypes = [v, i, v, v, v, v, v, v, i, i, v, v, v, v, v, v]
    hip.solr_hip_probe_postprocess.argtypes = [v, v, v, v]
    hip.solr_hip_h2d_randoms_sized.argtypes = [v, C.c_long]
    hip.solr_hip_probe_ticket.argtypes = [C.c_longlong, v, v]
    hip.solr_hip_probe_image_serial.argtypes = [C.c_longlong]
    hip.solr_hip_probe_image_serial.restype = C.c_longlong
    hip.solr_hip_probe_vectors.argtypes = [i, v, v, v, v, v, v]
    hip.solr_hip_probe_make_color.argtypes = [v, i, v, v]
    hip.solr_hip_probe_skybox.argtypes = [v, i, v, v, v]
    hip.solr_hip_probe_intersection_shader.argtypes = [v, i, v, v, v, v, v, v, v]
    hip.h2d_textures.argtypes = [C.c_uint64, i, v]
    for name in ("box", "box_walk", "primitive", "closest", "shadow", "shader", "postprocess", "ticket", "vectors",
                 "make_color", "skybox", "intersection_shader"):
        getattr(hip, "solr_hip_probe_" + name).restype = i
    del P


def _check(hip, status, what):
    buf = C.create_string_buffer(512)
    code = hip.solr_hip_last_error(buf, 512)
    if status < 0 or code != 0:
        hip.solr_hip_clear_error()
        raise RuntimeError("%s failed (status %d, error %d): %s" % (what, status, code, buf.value.decode(errors="replace")))
    return status


class Resident:
    """a scene made resident through the C ABI alone; `with Resident(...) as r:` finalizes the engine afterwards"""

    def __init__(self, solr, si, boxes, prims, materials, textures=None, lights=None, nb_lamps=0, randoms=None):
        self.solr, self.hip = solr, solr.hip_lib()
        declare(self.hip)
        hip = self.hip
        hip.solr_hip_clear_error()
        hip.solr_hip_set_variant(0)
        hip.solr_hip_initialize(C.byref(si))
        used = np.flatnonzero(np.frombuffer(materials.tobytes(), np.uint8).reshape(len(materials), -1).any(axis=1))
        nb = int(used.max()) + 1 if len(used) else 1
        self.keep = [np.ascontiguousarray(materials[:nb]), np.ascontiguousarray(boxes), np.ascontiguousarray(prims)]
        hip.h2d_materials(0, _p(self.keep[0]), nb)
        if textures is not None and len(textures):
            tex = np.ascontiguousarray(textures)
            info = TextureInfo(tex.ctypes.data, 0, (C.c_int * 3)(len(tex), 1, 1), 0, 0)
            self.keep += [tex, info]
            hip.h2d_textures(0, 1, C.addressof(info))
        lamps = np.zeros(max(nb_lamps, 1), i32)
        self.keep.append(lamps)
        hip.h2d_scene(0, _p(self.keep[1]), len(boxes), _p(self.keep[2]), len(prims), _p(lamps), nb_lamps)
        if lights is not None and len(lights):
            li = np.ascontiguousarray(lights)
            self.keep.append(li)
            hip.h2d_lightInformation(0, _p(li), len(li))
        if randoms is not None and len(randoms):
            rnd = np.ascontiguousarray(randoms, f32)
            self.keep.append(rnd)
            hip.solr_hip_h2d_randoms_sized(_p(rnd), len(rnd))
        _check(hip, 0, "upload")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.hip.finalize_scene(0)
        self.hip.solr_hip_clear_error()
        return False


def one_leaf(solr, prims):
    """a node list of one leaf that holds every primitive (the per-primitive probes never walk it)"""
    boxes = np.zeros(1, solr.BOX_DTYPE)
    boxes["min"], boxes["max"] = -1.0e6, 1.0e6
    boxes["nbPrimitives"], boxes["startIndex"] = len(prims), 0
    boxes["indexForNextBox"][:, 0] = 1
    return boxes


def engine_outputs(solr, case, features=0, exact=0):
    """the case evaluated by the engine's device functions; keys as oracle.probes.oracle_outputs"""
    hip = solr.hip_lib()
    declare(hip)
    name = case["name"]
    si = case.get("si")
    out = {}
    if name == "box":
        n = len(case["origins"])
        exact_form, fast_form = np.zeros(n, i32), np.zeros(n, i32)
        _check(hip, hip.solr_hip_probe_box(n, _p(case["boxes"]), _p(case["origins"]), _p(case["directions"]), _p(case["t0"]),
                                           _p(case["t1"]), _p(exact_form), _p(fast_form)), "probe_box")
        out = dict(hit=exact_form, hit_fast=fast_form)
        # the node loop: the same boxes as a flat list of leaves, one dummy primitive each
        boxes = np.array(case["boxes"], copy=True)
        boxes["nbPrimitives"], boxes["startIndex"] = 1, np.arange(n)
        boxes["indexForNextBox"][:, 0] = 1
        boxes["indexForNextBox"][:, 1] = 0
        prims = np.zeros(n, solr.PRIMITIVE_DTYPE)
        prims["size"][:, 0] = 1.0
        prims["index"] = np.arange(n)
        materials = np.zeros(2, solr.MATERIAL_DTYPE)
        materials["color"][:, :3] = 0.5
        from oracle import probes
        with Resident(solr, probes._scene_info(), boxes, prims, materials):
            for key, feat in (("hit_walk", F_SPHERE | F_PLANE), ("hit_walk_deep", F_SPHERE | F_PLANE | F_DEEP)):
                walked = np.zeros(n, i32)
                _check(hip, hip.solr_hip_probe_box_walk(C.byref(probes._scene_info()), n, _p(case["origins"]),
                                                        _p(case["directions"]), _p(case["t1"]), feat, _p(walked)), key)
                out[key] = walked
        return out
    if name == "primitive":
        n = len(case["origins"])
        inter = np.ascontiguousarray(case["initial"][:, 0, :]).copy()
        normal = np.ascontiguousarray(case["initial"][:, 1, :]).copy()
        areas, shadow, hit = np.zeros((n, 3), f32), np.zeros(n, f32), np.zeros(n, i32)
        with Resident(solr, si, one_leaf(solr, case["prims"]), case["prims"], case["materials"], case["textures"]):
            used = _check(hip, hip.solr_hip_probe_primitive(C.byref(si), n, _p(case["origins"]), _p(case["directions"]),
                                                            _p(case["shadows"]), features, _p(inter), _p(normal), _p(areas),
                                                            _p(shadow), _p(hit)), "probe_primitive")
        return dict(hit=hit, intersection=inter, normal=normal, areas=areas, shadow=shadow, features=used)
    if name in ("closest", "shadow"):
        s = case["scene"]
        n = len(case["origins"])
        with Resident(solr, si, s.boxes, s.prims, s.materials, s.textures, s.lights, s.nb_lamps):
            if name == "closest":
                hit, prim = np.zeros(n, i32), np.zeros(n, i32)
                inter, normal, areas = np.zeros((n, 3), f32), np.zeros((n, 3), f32), np.zeros((n, 3), f32)
                used = _check(hip, hip.solr_hip_probe_closest(C.byref(si), n, _p(case["origins"]), _p(case["targets"]),
                                                              _p(case["iteration"]), _p(case["current"]), features, exact,
                                                              _p(hit), _p(prim), _p(inter), _p(normal), _p(areas)),
                              "probe_closest")
                return dict(hit=hit, primitive=prim, intersection=inter, normal=normal, areas=areas, features=used)
            result, color = np.zeros(n, f32), np.zeros((n, 3), f32)
            # the reference's probe (the OpenCL engine) leaves out the lamp only; the renderer's call also the primitive
            # the point lies on (GI:829) - the cases that name it (`shaded`) are probed the renderer's way
            nobody = case["shaded"] if "shaded" in case else np.full(n, NOBODY, i32)
            used = _check(hip, hip.solr_hip_probe_shadow(C.byref(si), n, _p(case["lamps"]), _p(case["origins"]),
                                                         _p(case["object_id"]), _p(nobody), _p(case["iteration"]), features,
                                                         exact, _p(result), _p(color)), "probe_shadow")
            return dict(result=result, color=color, features=used)
    if name == "shader":
        s = case["scene"]
        n = len(case["origins"])
        normal, cc, tb = case["normal"].copy(), case["closest_color"].copy(), case["total_blinn"].copy()
        at = case["attributes"].copy()
        ret, shadow = np.zeros((n, 3), f32), np.zeros(n, f32)
        with Resident(solr, si, s.boxes, s.prims, s.materials, s.textures, s.lights, s.nb_lamps, s.randoms):
            used = _check(hip, hip.solr_hip_probe_shader(C.byref(si), n, _p(case["index"]), _p(case["origins"]),
                                                         _p(case["object_id"]), _p(case["inter"]), _p(case["areas"]),
                                                         _p(case["iteration"]), features, exact, _p(normal), _p(cc), _p(tb),
                                                         _p(at), _p(ret), _p(shadow)), "probe_shader")
        return dict(returned=ret, shadow=shadow, normal=normal, closest_color=cc, total_blinn=tb, attributes=at,
                    features=used)
    if name == "launch":
        # the case's rays are the camera rays of k_standardRenderer for this eye / look-at point and no rotation
        # (oracle.probes.case_launch): the frame goes through the renderer itself, pass 0
        s = case["scene"]
        w, h = case["width"], case["height"]
        eye, look = case["origins"][0].copy(), np.array([57.0, 23.0, 0.0], f32)
        angles = np.array([0.0, 0.0, 0.0, 6400.0], f32)
        assert (case["origins"] == eye).all() and case["targets"][(h // 2) * w + w // 2].tolist() == look.tolist()
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        with Resident(solr, si, s.boxes, s.prims, s.materials, s.textures, s.lights, s.nb_lamps, s.randoms):
            hip.solr_hip_set_variant(4 if features == EVERYTHING else (3 if exact else 0))
            try:
                objects = solr.Vec4i(len(s.boxes), len(s.prims), s.nb_lamps, len(s.lights))
                hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(case["ppi"]), fp(eye), fp(look), fp(angles))
                _check(hip, 0, "render")
                pp = np.zeros((h, w, 8), f32)
                ids = np.zeros((h, w, 4), i32)
                rgb = np.zeros((h, w, 3), np.uint8)
                hip.solr_hip_d2h_postprocessing(C.c_void_p(pp.ctypes.data))
                hip.solr_hip_d2h(C.byref(si), C.c_void_p(rgb.ctypes.data), C.c_void_p(ids.ctypes.data))
                _check(hip, 0, "read-back")
            finally:
                hip.solr_hip_set_variant(0)
        pp = pp.reshape(-1, 8)
        return dict(color=pp[:, :3].copy(), depth=pp[:, 3].copy(), ids=ids.reshape(-1, 4).copy(), bitmap=rgb.reshape(-1))
    if name == "post":
        w, h = case["width"], case["height"]
        bitmap = np.zeros(w * h * 3, np.uint8)
        prims = np.zeros(1, solr.PRIMITIVE_DTYPE)
        prims["size"][:, 0] = 1.0
        materials = np.zeros(2, solr.MATERIAL_DTYPE)
        randoms = np.zeros(max(len(case["randoms"]), MAX_BITMAP_SIZE), f32)     # (the upload wants the reference's size)
        randoms[: len(case["randoms"])] = case["randoms"]
        with Resident(solr, si, one_leaf(solr, prims), prims, materials, randoms=randoms):
            _check(hip, hip.solr_hip_probe_postprocess(C.byref(si), C.byref(case["ppi"]), _p(case["pp"]), _p(bitmap)),
                   "probe_postprocess")
        return dict(bitmap=bitmap)
    if name == "vectors":
        n = len(case["incident"])
        refracted, reflected = np.zeros((n, 3), f32), np.zeros((n, 3), f32)
        _check(hip, hip.solr_hip_probe_vectors(n, _p(case["incident"]), _p(case["normals"]), _p(case["n1"]), _p(case["n2"]),
                                               _p(refracted), _p(reflected)), "probe_vectors")
        return dict(refracted=refracted, reflected=reflected)
    if name == "make_color":
        n = len(case["colors"])
        bitmap = np.zeros(n * 3, np.uint8)
        _check(hip, hip.solr_hip_probe_make_color(C.byref(si), n, _p(case["colors"]), _p(bitmap)), "probe_make_color")
        return dict(bitmap=bitmap)
    if name == "skybox":
        n = len(case["origins"])
        color = np.zeros((n, 3), f32)
        prims = np.zeros(1, solr.PRIMITIVE_DTYPE)
        prims["size"][:, 0] = 1.0
        with Resident(solr, si, one_leaf(solr, prims), prims, case["materials"], case["textures"]):
            _check(hip, hip.solr_hip_probe_skybox(C.byref(si), n, _p(case["origins"]), _p(case["targets"]), _p(color)),
                   "probe_skybox")
        return dict(color=color)
    if name == "intersection_shader":
        n = len(case["inter"])
        at = case["attributes"].copy()
        color, bump, spec, ao = np.zeros((n, 4), f32), np.zeros((n, 3), f32), np.zeros((n, 3), f32), np.zeros((n, 1), f32)
        with Resident(solr, si, one_leaf(solr, case["prims"]), case["prims"], case["materials"], case["textures"]):
            _check(hip, hip.solr_hip_probe_intersection_shader(C.byref(si), n, _p(case["inter"]), _p(case["areas"]), _p(at),
                                                               _p(color), _p(bump), _p(spec), _p(ao)),
                   "probe_intersection_shader")
        return dict(color=color, bump=bump, specular=spec, advanced=ao, attributes=at)
    raise KeyError(name)


ENGINE_CASES = ("box", "primitive", "closest", "shadow", "shader", "launch", "post", "vectors", "make_color", "skybox",
                "intersection_shader")
