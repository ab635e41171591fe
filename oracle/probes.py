"""Function-level pin of the CPU oracle against the reference's own functions.

TEST INFRASTRUCTURE ONLY (tests/test_reference_probes.py, tests/golden/make_probe_fixtures.py).

oracle/ref_probes.cl wraps the functions of the reference's OpenCL engine (RayTracer.cl, #included from
/root/reference at build time) in kernels that evaluate them over arrays of inputs; this module
  * builds those arrays, deterministically from seeds (`cases`),
  * runs the probes on the GPU through oracle/_ref (`reference_outputs`),
  * runs the same evaluations through oracle/libsolr_oracle.so in its OpenCL dialect (`oracle_outputs`;
    solr_oracle.c, "dialect": the statements in which the reference's two engines differ, each an
    `if (g_cl)` citing both files - everything else is shared with the CUDA dialect the product matches),
  * and says how the two must agree (`compare`): bit for bit against the source-order build of the probes,
    within a stated bound against the build with ROCm's fused / approximate geometric builtins.
"""
import ctypes as C
import hashlib
import importlib
import math
import os

import numpy as np

from . import loader

_HERE = os.path.dirname(os.path.abspath(__file__))
PROBES = {"as_built": os.path.join(_HERE, "_ref", "ref_probes_gfx950.co"),
          "source_order": os.path.join(_HERE, "_ref", "ref_probes_srcorder_gfx950.co")}
RENDERER = os.path.join(_HERE, "_ref", "RayTracer_gfx950.co")

f32, i32 = np.float32, np.int32
RANDOMS_SHA1 = "371b6e06a3b5d8e2f3672b076812e15bfd49cef9"   # sha1 of the host's random buffer for seed 1
CL_BOX = np.dtype({"names": ["min", "max", "nbPrimitives", "startIndex", "indexForNextBox"],
                   "formats": [(f32, 4), (f32, 4), i32, i32, (i32, 2)], "offsets": [0, 16, 32, 36, 40], "itemsize": 48})
CL_PRIM = np.dtype({"names": ["p0", "p1", "p2", "n0", "n1", "n2", "size", "type", "index", "materialId", "vt0", "vt1",
                              "vt2"],
                    "formats": [(f32, 4)] * 7 + [i32, i32, i32] + [(f32, 2)] * 3,
                    "offsets": [0, 16, 32, 48, 64, 80, 96, 112, 116, 120, 128, 136, 144], "itemsize": 160})
CL_LIGHT = np.dtype({"names": ["primitiveId", "materialId", "location", "color"],
                     "formats": [i32, i32, (f32, 4), (f32, 4)], "offsets": [0, 4, 16, 32], "itemsize": 48})


def have_probes():
    return loader.have_ref() and all(os.path.exists(p) for p in PROBES.values())


def f4(a):
    """(n, 3) -> (n, 4) with w = 0 (the OpenCL engine's vectors)"""
    a = np.asarray(a, f32).reshape(-1, 3)
    out = np.zeros((len(a), 4), f32)
    out[:, :3] = a
    return out


def cl_boxes(boxes):
    out = np.zeros(len(boxes), CL_BOX)
    out["min"][:, :3] = boxes["min"]
    out["max"][:, :3] = boxes["max"]
    for k in ("nbPrimitives", "startIndex", "indexForNextBox"):
        out[k] = boxes[k]
    return out


def cl_prims(prims):
    out = np.zeros(len(prims), CL_PRIM)
    for k in ("p0", "p1", "p2", "n0", "n1", "n2", "size"):
        out[k][:, :3] = prims[k]
    for k in ("type", "index", "materialId", "vt0", "vt1", "vt2"):
        out[k] = prims[k]
    return out


def cl_lights(lights):
    out = np.zeros(max(len(lights), 1), CL_LIGHT)
    n = len(lights)
    out["primitiveId"][:n] = lights["primitiveId"]
    out["materialId"][:n] = lights["materialId"]
    out["location"][:n, :3] = lights["location"]
    out["color"][:n] = lights["color"]
    return out


# ---- the reference side --------------------------------------------------------------------------------
class _RefArg(C.Structure):
    _fields_ = [("kind", C.c_int), ("data", C.c_void_p), ("bytes", C.c_size_t)]


def val(x):
    """a by-value kernel argument: a ctypes structure / scalar or raw bytes"""
    return ("val", x)


def run_kernel(code_object, kernel, args, global_size, local_size=None):
    """args: list of ("val", ctypes object) | ("in" | "out" | "inout", ndarray); out / inout arrays are
    overwritten in place."""
    R = C.CDLL(loader.REF_LIB)
    R.solr_ref_opencl_run.restype = C.c_int
    keep, arr = [], (_RefArg * len(args))()
    for i, (kind, obj) in enumerate(args):
        if kind == "val":
            if isinstance(obj, (int, np.integer)):
                obj = C.c_int(int(obj))
            elif isinstance(obj, float):
                obj = C.c_float(obj)
            keep.append(obj)
            arr[i] = _RefArg(0, C.addressof(obj), C.sizeof(obj))
        else:
            assert isinstance(obj, np.ndarray) and obj.flags["C_CONTIGUOUS"], (kernel, i)
            keep.append(obj)
            arr[i] = _RefArg({"in": 1, "out": 2, "inout": 3}[kind], obj.ctypes.data, obj.nbytes)
    dims = len(global_size)
    g = (C.c_size_t * dims)(*global_size)
    l = (C.c_size_t * dims)(*local_size) if local_size else None
    log = C.create_string_buffer(4096)
    status = R.solr_ref_opencl_run(code_object.encode(), kernel.encode(), C.c_int(len(args)), arr, C.c_int(dims), g, l,
                                   log, C.c_int(4096))
    if status != 0:
        raise RuntimeError("reference probe %s failed (%d): %s" % (kernel, status, log.value.decode(errors="replace")))


# ---- scenes ----------------------------------------------------------------------------------------------
def _solr():
    return importlib.import_module("sol-r_amd")


class SceneData:
    """flattened arrays of a scene built through the host mirror (host-only engine: no GPU involved)"""

    def __init__(self, build, **kw):
        solr = _solr()
        k = solr.Kernel(engine="host-only", deterministic_seed=1)
        build(k, **kw)
        flat = k.flat_scene()
        self.si, self.ppi, self.eye, self.direction, self.angles = k.frame_parameters()
        self.boxes = np.array(flat.boxes, copy=True)
        self.prims = np.array(flat.primitives, copy=True)
        self.lights = np.array(flat.lights, copy=True)
        self.nb_lamps = flat.nb_lamps
        self.textures = np.array(flat.textures, copy=True)
        self.randoms = np.array(flat.randoms, copy=True)
        mats = np.array(flat.materials, copy=True)
        capacity = 65506 + 30 + 1     # NB_MAX_MATERIALS + 1, as loader.Scene pads it
        full = np.zeros(capacity, mats.dtype)
        full[: len(mats)] = mats
        self.materials = full
        if len(self.textures):
            self.textures = np.concatenate([self.textures, np.zeros(len(self.textures) + 4, np.uint8)])
        else:
            self.textures = np.zeros(16, np.uint8)
        k.finalize()

    @classmethod
    def from_arrays(cls, d):
        self = cls.__new__(cls)
        solr = _solr()
        self.si, self.ppi = solr.SceneInfo(), solr.PostProcessingInfo()
        C.memmove(C.addressof(self.si), d["si"].tobytes(), C.sizeof(self.si))
        C.memmove(C.addressof(self.ppi), d["ppi"].tobytes(), C.sizeof(self.ppi))
        self.eye, self.direction, self.angles = d["eye"], d["direction"], d["angles"]
        self.boxes = np.ascontiguousarray(d["boxes"]).view(solr.BOX_DTYPE).reshape(-1)
        self.prims = np.ascontiguousarray(d["prims"]).view(solr.PRIMITIVE_DTYPE).reshape(-1)
        self.lights = np.ascontiguousarray(d["lights"]).view(solr.LIGHT_DTYPE).reshape(-1)
        self.nb_lamps = int(d["nb_lamps"])
        self.textures = np.ascontiguousarray(d["textures"])
        mats = np.ascontiguousarray(d["materials"]).view(solr.MATERIAL_DTYPE).reshape(-1)
        full = np.zeros(65506 + 30 + 1, solr.MATERIAL_DTYPE)
        full[: len(mats)] = mats
        self.materials = full
        self.randoms = default_randoms() if int(d["has_randoms"]) else np.zeros(0, f32)
        return self

    def to_arrays(self):
        """what from_arrays needs, as plain byte / float arrays (the records' padding zeroed)"""
        used = np.flatnonzero(np.frombuffer(self.materials.tobytes(), np.uint8).reshape(len(self.materials), -1).any(axis=1))
        nb = int(used.max()) + 1 if len(used) else 1
        return dict(si=np.frombuffer(bytes(self.si), np.uint8), ppi=np.frombuffer(bytes(self.ppi), np.uint8),
                    eye=self.eye, direction=self.direction, angles=self.angles, boxes=_clean_bytes(self.boxes),
                    prims=_clean_bytes(self.prims), lights=_clean_bytes(self.lights), nb_lamps=np.int32(self.nb_lamps),
                    textures=self.textures, materials=_clean_bytes(self.materials[:nb]),
                    has_randoms=np.int32(1 if len(self.randoms) else 0))

    def oracle_scene(self):
        rnd = self.randoms if len(self.randoms) else np.zeros(1, f32)
        self._keep = rnd
        return loader.OracleScene(self.boxes.ctypes.data, len(self.boxes), self.prims.ctypes.data, len(self.prims),
                                  self.lights.ctypes.data, len(self.lights), self.nb_lamps, self.materials.ctypes.data,
                                  self.textures.ctypes.data, rnd.ctypes.data, len(rnd))


def _clean_bytes(records):
    """the records as bytes with the never-written padding between / after the named fields zeroed"""
    clean = np.zeros(len(records), records.dtype)
    for f in records.dtype.names:
        clean[f] = records[f]
    return np.frombuffer(clean.tobytes(), np.uint8)


_RANDOMS = None


def default_randoms():
    """the host's random buffer for seed 1 (an integer generator times a constant: the same on every machine)"""
    global _RANDOMS
    if _RANDOMS is None:
        _RANDOMS = SceneData(_extra().lone_light).randoms
        assert hashlib.sha1(_RANDOMS.tobytes()).hexdigest() == RANDOMS_SHA1, "the host's random buffer changed"
    return _RANDOMS


def pack(case):
    """a case as a flat dict of arrays (what the fixture stores)"""
    out = {}
    for key, v in case.items():
        if isinstance(v, str):
            out["name"] = np.frombuffer(v.encode(), np.uint8)
        elif isinstance(v, SceneData):
            for k2, a in v.to_arrays().items():
                out["scene." + k2] = np.asarray(a)
        elif isinstance(v, C.Structure):
            out["struct." + key] = np.frombuffer(bytes(v), np.uint8)
        elif isinstance(v, np.ndarray) and v.dtype.names:
            if key == "materials":   # NB_MAX_MATERIALS + 1 records, zeros beyond the ones a scene defined
                used = np.flatnonzero(np.frombuffer(v.tobytes(), np.uint8).reshape(len(v), -1).any(axis=1))
                v = v[: (int(used.max()) + 1 if len(used) else 1)]
            out["records." + key] = _clean_bytes(v)
        elif isinstance(v, np.ndarray):
            out["array." + key] = v
        else:
            out["int." + key] = np.int64(v)
    return out


def unpack(d):
    """the inverse of pack"""
    solr = _solr()
    record_types = {"boxes": solr.BOX_DTYPE, "prims": solr.PRIMITIVE_DTYPE, "pp": solr.PP_DTYPE}
    case, scene = {}, {}
    for key, v in d.items():
        kind, _, rest = key.partition(".")
        if key == "name":
            case["name"] = bytes(v).decode()
        elif kind == "scene":
            scene[rest] = v
        elif kind == "struct":
            obj = solr.PostProcessingInfo() if rest == "ppi" else solr.SceneInfo()
            C.memmove(C.addressof(obj), np.ascontiguousarray(v).tobytes(), C.sizeof(obj))
            case[rest] = obj
        elif kind == "records":
            if rest == "materials":
                mats = np.ascontiguousarray(v).view(solr.MATERIAL_DTYPE).reshape(-1)
                full = np.zeros(65506 + 30 + 1, solr.MATERIAL_DTYPE)
                full[: len(mats)] = mats
                case[rest] = full
            else:
                case[rest] = np.ascontiguousarray(v).view(record_types[rest]).reshape(-1).copy()
        elif kind == "array":
            case[rest] = np.ascontiguousarray(v)
        elif kind == "int":
            case[rest] = int(v)
    if scene:
        case["scene"] = SceneData.from_arrays(scene)
    return case


def _scene_info(**kw):
    solr = _solr()
    si = solr.SceneInfo()
    base = dict(size_x=64, size_y=64, cameraType=0, graphicsLevel=4, nbRayIterations=3, transparentColor=2.0,
                viewDistance=50000.0, shadowIntensity=1.0, eyeSeparation=0.0, renderBoxes=0, pathTracingIteration=0,
                maxPathTracingIterations=100, frameBufferType=0, timestamp=0, atmosphericEffect=0,
                doubleSidedTriangles=0, extendedGeometry=1, advancedIllumination=0, draftMode=0, skyboxRadius=0,
                skyboxMaterialId=-1, gradientBackground=0, geometryEpsilon=0.001, rayEpsilon=0.05)
    base.update(kw)
    for k, v in base.items():
        setattr(si, k, v)
    for i, v in enumerate(base.get("backgroundColor", (0.0, 0.0, 0.0, 0.5))):
        si.backgroundColor[i] = v
    return si


def _copy_si(si, **kw):
    solr = _solr()
    out = solr.SceneInfo()
    C.memmove(C.addressof(out), C.addressof(si), C.sizeof(si))
    for k, v in kw.items():
        setattr(out, k, v)
    return out


def _unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f32)


# ---- cases -----------------------------------------------------------------------------------------------
def case_box(seed=11, n=2048):
    rng = np.random.default_rng(seed)
    lo = rng.uniform(-10000, 9000, (n, 3)).astype(f32)
    ext = rng.uniform(1.0, 6000, (n, 3)).astype(f32)
    solr = _solr()
    boxes = np.zeros(n, solr.BOX_DTYPE)
    boxes["min"], boxes["max"] = lo, lo + ext
    origins = rng.uniform(-15000, 15000, (n, 3)).astype(f32)
    centre = lo + ext * rng.uniform(-0.3, 1.3, (n, 3)).astype(f32)
    directions = ((centre - origins) * rng.uniform(0.2, 3.0, (n, 1))).astype(f32)
    axis = rng.integers(0, 4, n)           # a quarter of the rays run along an axis plane: zero components
    for a in range(3):
        directions[axis == a, a] = 0.0
    inside = rng.random(n) < 0.1           # origins inside the box
    origins[inside] = (lo + ext * 0.5)[inside]
    t0 = np.where(rng.random(n) < 0.5, 0.0, 0.05).astype(f32)
    t1 = rng.uniform(0.3, 2.0, n).astype(f32)
    return dict(name="box", boxes=boxes, origins=origins, directions=directions, t0=t0, t1=t1)


def _primitive_zoo(k, **info):
    """one of each primitive record the intersection routines read, derived by the host's setPrimitive"""
    solr = _solr()
    k.initialize(width=64, height=64, nbRayIterations=1, **info)
    plain = k.add_material(0.7, 0.3, 0.2, specValue=0.5, specPower=50.0)
    glass = k.add_material(0.9, 0.95, 1.0, reflection=0.8, refraction=1.2, transparency=0.6, opacity=0.2)
    bright = k.add_material(0.9, 0.9, 0.9)
    k.add_primitive(solr.ptSphere, (-3000, 500, 0), size=(1500, 0, 0), material=plain)
    k.add_primitive(solr.ptSphere, (800, -500, -2500), size=(1100, 0, 0), material=glass)
    k.add_primitive(solr.ptSphere, (10, 20, 30), size=(0.5, 0, 0), material=plain)
    k.add_primitive(solr.ptEllipsoid, (0, -2500, 1000), size=(2200, 700, 1200), material=plain)
    k.add_primitive(solr.ptEllipsoid, (2000, 2500, -1000), size=(300, 900, 500), material=glass)
    k.add_primitive(solr.ptCylinder, (-4500, -3000, -1000), (-2500, 2500, 500), size=(350, 0, 0), material=plain)
    k.add_primitive(solr.ptCylinder, (1000, 1000, 1000), (1000, 4000, 1000), size=(200, 0, 0), material=plain)
    k.add_primitive(solr.ptCylinder, (4000, 0, 0), (4100, 50, 30), size=(20, 0, 0), material=glass)
    for i in range(6):
        a0, a1 = 0.6 * i, 0.6 * (i + 1)
        p0 = (0.0, 3500.0, 2500.0)
        p1 = (2500.0 * math.cos(a0), 3500.0 + 900.0 * math.sin(3 * a0), 2500.0 + 2500.0 * math.sin(a0))
        p2 = (2500.0 * math.cos(a1), 3500.0 + 900.0 * math.sin(3 * a1), 2500.0 + 2500.0 * math.sin(a1))
        t = k.add_primitive(solr.ptTriangle, p0, p1, p2, material=plain if i % 2 else glass)
        k.set_normals(t, (0.1, -1, 0.2), (0.3 * math.cos(a0), -1, 0.3 * math.sin(a0)),
                      (0.3 * math.cos(a1), -1, 0.3 * math.sin(a1)))
    k.add_primitive(solr.ptXYPlane, (0, 0, 9000), size=(9000, 6000, 0), material=plain)
    k.add_primitive(solr.ptYZPlane, (-9000, 0, 3000), size=(0, 6000, 6000), material=plain)
    k.add_primitive(solr.ptYZPlane, (9000, 0, 3000), size=(0, 6000, 6000), material=bright)
    k.add_primitive(solr.ptXZPlane, (0, 6000, 3000), size=(9000, 0, 6000), material=plain)
    k.add_primitive(solr.ptXZPlane, (0, -6000, 3000), size=(9000, 0, 6000), material=glass)
    k.add_primitive(solr.ptCheckboard, (0, -4500, 2000), size=(9000, 0, 7000), material=plain)
    # (no ptCamera: its plane test samples a texture whatever the material, RayTracer.cl:1290-1297, and an
    # untextured material carries the 40000 x 40000 "computed texture" mapping - a read far outside the atlas)
    lm = k.add_material(1.0, 1.0, 1.0, innerIllumination=2.0)
    k.add_primitive(solr.ptSphere, (8000, 8000, -8000), size=(10, 0, 0), material=lm)
    k.compact_boxes(True)
    k.set_camera((0.0, 0.0, -15000.0))


def _extent(p):
    """a radius that covers primitive record p around its p0 (for aiming rays at it)"""
    solr = _solr()
    t = int(p["type"])
    if t in (solr.ptSphere, solr.ptEnvironment):
        return float(p["size"][0])
    if t == solr.ptEllipsoid:
        return float(max(p["size"]))
    if t in (solr.ptCylinder, solr.ptCone):
        return float(np.linalg.norm(p["p1"] - p["p0"])) + float(p["size"][0])
    if t == solr.ptTriangle:
        return float(max(np.linalg.norm(p["p1"] - p["p0"]), np.linalg.norm(p["p2"] - p["p0"])))
    return float(max(p["size"]))


def case_primitive(seed=12, per_primitive=96, double_sided=0, extended=1, transparent_color=2.0):
    rng = np.random.default_rng(seed + 7 * double_sided + 13 * (1 - extended))
    scene = SceneData(_primitive_zoo)
    solr = _solr()
    sel = [i for i in range(len(scene.prims)) if extended or scene.prims["type"][i] == solr.ptTriangle]
    choice = np.repeat(np.array(sel, i32), per_primitive)
    n = len(choice)
    prims = scene.prims[choice].copy()
    origins = np.zeros((n, 3), f32)
    directions = np.zeros((n, 3), f32)
    for j in range(n):
        p = prims[j]
        r = _extent(p)
        c = p["p0"].astype(np.float64)
        if int(p["type"]) == solr.ptTriangle:
            c = (p["p0"] + p["p1"] + p["p2"]).astype(np.float64) / 3.0
        if int(p["type"]) in (solr.ptCylinder, solr.ptCone):
            c = (p["p0"] + p["p1"]).astype(np.float64) / 2.0
            r = r / 2.0
        mode = j % 8
        u = _unit(rng, 2).astype(np.float64)
        if mode == 0:      # from inside (back faces of spheres, t1 <= epsilon branches)
            o = c + u[0] * r * rng.uniform(0.0, 0.6)
        else:
            o = c + u[0] * r * rng.uniform(1.5, 12.0)
        if mode == 1:      # grazing: aim at the silhouette
            side = np.cross(u[0], u[1])
            side /= max(np.linalg.norm(side), 1e-9)
            target = c + side * r * (1.0 + rng.choice([-1e-5, 1e-5, -1e-3, 1e-3]))
        else:
            target = c + u[1] * r * rng.uniform(0.0, 1.4)
        d = (target - o) * rng.uniform(0.4, 3.0)   # unnormalised, as the walks hand it over
        if mode == 2:      # along an axis: zero components
            a = int(rng.integers(0, 3))
            d[(a + 1) % 3] = 0.0
            d[(a + 2) % 3] = 0.0
            d[a] = np.sign(c[a] - o[a] + 1e-9) * (abs(c[a] - o[a]) + r) * rng.uniform(0.5, 2.0)
            o[(a + 1) % 3] = c[(a + 1) % 3] + r * rng.uniform(-0.5, 0.5)
            o[(a + 2) % 3] = c[(a + 2) % 3] + r * rng.uniform(-0.5, 0.5)
        origins[j], directions[j] = o, d
    shadows = (rng.random(n) < 0.3).astype(i32)
    si = _scene_info(doubleSidedTriangles=double_sided, extendedGeometry=extended, transparentColor=transparent_color)
    initial = np.zeros((n, 2, 3), f32)     # the walk hands the routines whatever its locals held: start from zero
    return dict(name="primitive", si=si, prims=prims, materials=scene.materials, textures=scene.textures,
                origins=origins, directions=directions, shadows=shadows, initial=initial)


def _walk_rays(scene, rng, n):
    lo = scene.boxes["min"].min(axis=0)
    hi = scene.boxes["max"].max(axis=0)
    lo, hi = np.maximum(lo, -12000), np.minimum(hi, 12000)
    origins = np.empty((n, 3), f32)
    third = n // 3
    origins[:third] = scene.eye + rng.normal(size=(third, 3)) * 30.0
    origins[third:] = rng.uniform(lo, hi, (n - third, 3))
    targets = rng.uniform(lo, hi, (n, 3)).astype(f32)
    # a tenth of the rays aimed exactly at primitive anchor points (ties, box corners)
    pick = rng.integers(0, len(scene.prims), n // 10)
    targets[: n // 10] = scene.prims["p0"][pick]
    return origins.astype(f32), targets


SCENES = {
    "cornell": lambda: SceneData(_solr().scenes.cornell, width=128, height=96, iterations=3),
    # the same room without its glass spheres: one lamp, nothing transparent, nothing emissive but the lamp
    "cornell_opaque": lambda: SceneData(_solr().scenes.cornell, width=128, height=96, iterations=3, glass=0),
    "mix": lambda: SceneData(_extra().primitives_mix, width=96, height=64),
    "sticks": lambda: SceneData(_extra().sticks),
    "triangles": lambda: SceneData(_extra().triangles_only),
    "triangles_flat": lambda: SceneData(_extra().triangles_only, extendedGeometry=0),
    "textured": lambda: SceneData(_extra().textured, width=96, height=64),
    # BASELINE configs[2] / [3] in small, shaped so that none of the statements in which the reference's two engines
    # differ shows on (nearly) any pixel: one lamp, nothing transparent or emissive in view, a pass below 10, colours
    # inside [0, 1] (no final saturate), a backdrop behind the sticks (no ray misses: the depth word of a miss)
    "mesh_100": lambda: SceneData(_extra().triangles_only, n=7, dim=0.6, backdrop=True),
    "sticks_backdrop": lambda: SceneData(_extra().sticks, backdrop=True, dim=0.6),
    # every primitive type under ONE lamp (the lamp loop differs only with several)
    "mix_one_lamp": lambda: SceneData(_extra().primitives_mix, width=96, height=64, lamps=1),
}


def _extra():
    import sys
    tests = os.path.join(os.path.dirname(_HERE), "tests")
    if tests not in sys.path:
        sys.path.insert(0, tests)
    return importlib.import_module("scenes_extra")


def case_closest(scene_name, seed=13, n=1536):
    rng = np.random.default_rng(seed + len(scene_name))
    scene = SCENES[scene_name]()
    origins, targets = _walk_rays(scene, rng, n)
    iteration = rng.choice([0, 0, 1, 2, 3, 5, 9], n).astype(i32)
    mats = np.unique(scene.prims["materialId"])
    current = np.where(rng.random(n) < 0.5, -2, rng.choice(mats, n)).astype(i32)
    return dict(name="closest", scene=scene, si=scene.si, origins=origins, targets=targets, iteration=iteration,
                current=current)


def _surface_points(scene, rng, n):
    """points on surfaces with the data the shader is handed: found with the oracle's own closest-hit walk
    (inputs only; both sides are then given the same arrays)"""
    L = loader.lib()
    origins, targets = _walk_rays(scene, rng, 3 * n)
    m = len(origins)
    hit, prim = np.zeros(m, i32), np.zeros(m, i32)
    inter, normal, areas = np.zeros((m, 3), f32), np.zeros((m, 3), f32), np.zeros((m, 3), f32)
    osc = scene.oracle_scene()
    was = L.oracle_get_dialect()
    L.oracle_set_dialect(1)
    first_bounce, no_material = np.zeros(m, i32), np.full(m, -2, i32)   # (named: they must outlive the call)
    L.oracle_probe_closest(m, C.byref(osc), C.byref(scene.si), _p(origins), _p(targets), _p(first_bounce),
                           _p(no_material), _p(hit), _p(prim), _p(inter), _p(normal), _p(areas))
    L.oracle_set_dialect(was)
    keep = np.flatnonzero(hit)[:n]
    return origins[keep], prim[keep], inter[keep], normal[keep], areas[keep]


def case_shadow(scene_name, seed=14, n=1024, moot=False):
    """moot: the shadow rays of a scene in which the three statements that differ between the reference's two engines
    (processShadows: node test from 0 / 0.05, the shaded primitive left out / not, the transparent-shadow factor) cannot
    show - opaque occluders, and only the elements on which the oracle's two dialects return the same bits are kept
    (`_keep_where_the_dialects_agree`: input selection, like _surface_points).  Such a case also says which primitive
    each point lies on (`shaded`, its index in the flattened array): the CUDA engine's walk leaves that primitive out
    (GI:829), and the engine is probed the way its renderer calls it."""
    rng = np.random.default_rng(seed + len(scene_name) + (100 if moot else 0))
    scene = SCENES[scene_name]()
    origins, prim, inter, normal, areas = _surface_points(scene, rng, 2 * n if moot else n)
    n = len(inter) if not moot else n
    m = len(inter)
    light = scene.lights[0]
    lamp = (light["location"][None, :] + rng.normal(size=(m, 3)) * 40.0).astype(f32)
    far = rng.random(m) < 0.15            # lamps elsewhere: rays that cross the scene
    lamp[far] = rng.uniform(-9000, 9000, (int(far.sum()), 3)).astype(f32)
    object_id = np.full(m, int(light["primitiveId"]), i32)
    iteration = rng.choice([0, 1, 2, 3], m).astype(i32)
    case = dict(name="shadow", scene=scene, si=scene.si, lamps=lamp, origins=inter.copy(), object_id=object_id,
                iteration=iteration)
    if moot:
        case["shaded"] = prim.astype(i32)
        case = _keep_where_the_dialects_agree(case, n)
    return case


def case_shader(scene_name, seed=15, n=1024, moot=False, **si_changes):
    """moot: as case_shadow - a scene with one lamp, no emissive or wireframe material but the lamp's, a refinement pass
    below 10, and only elements on which the oracle's two dialects agree in every output: there the reference's own
    primitiveShader IS the CUDA engine's answer (statements 18-21 and, through its shadow rays, 14-16 do not show)."""
    rng = np.random.default_rng(seed + len(scene_name) + (100 if moot else 0))
    scene = SCENES[scene_name]()
    origins, prim, inter, normal, areas = _surface_points(scene, rng, 2 * n if moot else n)
    keep, n = n, len(inter)
    mats = scene.materials[scene.prims["materialId"][prim]]
    attributes = np.stack([mats["reflection"], mats["transparency"], mats["refraction"], mats["opacity"]], 1).astype(f32)
    closest_color = np.where(rng.random((n, 1)) < 0.5, 0.0, rng.uniform(0, 1, (n, 3))).astype(f32)
    total_blinn = np.where(rng.random((n, 1)) < 0.5, 0.0, rng.uniform(0, 0.3, (n, 3))).astype(f32)
    iteration = rng.choice([0, 1, 2, 3, 4, 5], n).astype(i32)
    index = rng.integers(0, 1920 * 1080, n).astype(i32)
    si = _copy_si(scene.si, **si_changes)
    case = dict(name="shader", scene=scene, si=si, ppi=scene.ppi, index=index, origins=origins, normal=normal,
                object_id=prim.astype(i32), inter=inter, areas=areas, closest_color=closest_color, iteration=iteration,
                total_blinn=total_blinn, attributes=attributes)
    return _keep_where_the_dialects_agree(case, keep) if moot else case


def case_intersection_shader(seed=16, per_primitive=120):
    rng = np.random.default_rng(seed)
    scene = SCENES["textured"]()
    solr = _solr()
    # fractal materials are left out: the CUDA engine iterates the Mandelbrot set with a binary64 step
    # (TextureMapping.cuh:172-175), the OpenCL engine in binary32 (RayTracer.cl:476-477)
    mat_ids = scene.materials["textureIds"][scene.prims["materialId"]][:, 0]
    sel = [i for i in range(len(scene.prims)) if mat_ids[i] >= 0]
    choice = np.repeat(np.array(sel, i32), per_primitive)
    n = len(choice)
    prims = scene.prims[choice].copy()
    inter = np.zeros((n, 3), f32)
    areas = np.zeros((n, 3), f32)
    for j in range(n):
        p = prims[j]
        t = int(p["type"])
        if t == solr.ptSphere:
            inter[j] = p["p0"] + _unit(rng, 1)[0] * p["size"][0]
        elif t == solr.ptTriangle:
            w = rng.dirichlet([1, 1, 1])
            inter[j] = w[0] * p["p0"] + w[1] * p["p1"] + w[2] * p["p2"]
            areas[j] = (w * rng.uniform(10, 1e6)).astype(f32)
        else:
            inter[j] = p["p0"] + rng.uniform(-1, 1, 3) * p["size"]
    mats = scene.materials[prims["materialId"]]
    attributes = np.stack([mats["reflection"], mats["transparency"], mats["refraction"], mats["opacity"]], 1).astype(f32)
    return dict(name="intersection_shader", si=_copy_si(scene.si, timestamp=3), prims=prims, materials=scene.materials,
                textures=scene.textures, inter=inter, areas=areas, attributes=attributes)


def case_skybox(seed=17, n=1024):
    rng = np.random.default_rng(seed)
    scene = SCENES["textured"]()
    origins = rng.uniform(-9000, 9000, (n, 3)).astype(f32)
    targets = (origins + _unit(rng, n) * rng.uniform(10, 20000, (n, 1))).astype(f32)
    outside = rng.random(n) < 0.1
    origins[outside] *= 8.0               # beyond the sky sphere
    return dict(name="skybox", si=scene.si, materials=scene.materials, textures=scene.textures, origins=origins,
                targets=targets)


def case_vectors(seed=18, n=1024):
    rng = np.random.default_rng(seed)
    incident = _unit(rng, n)
    normals = _unit(rng, n)
    n1 = rng.choice([1.0, 1.0, 1.1, 1.33, 1.5, 2.4], n).astype(f32)
    n2 = rng.choice([1.0, 1.0, 1.1, 1.33, 1.5, 2.4, 0.0], n).astype(f32)
    return dict(name="vectors", incident=incident, normals=normals, n1=n1, n2=n2)


def case_make_color(seed=19, frame_buffer_type=0):
    rng = np.random.default_rng(seed)
    n = 64 * 64
    colors = rng.uniform(-0.3, 1.3, (n, 3)).astype(f32)
    colors[:256] = (np.arange(256)[:, None] / 255.0).astype(f32)
    colors[256:512] = np.nextafter((np.arange(256)[:, None] / 255.0).astype(f32), f32(2.0))
    return dict(name="make_color", si=_scene_info(frameBufferType=frame_buffer_type), colors=colors)


def case_launch(scene_name, seed=20, width=80, height=56, **si_changes):
    """the camera rays of a width x height frame as k_standardRenderer sets them up for an unrotated camera
    (RayTracer.cl:2507-2512 / CudaRayTracer.cu:487-497), without the jitter either engine adds"""
    scene = SCENES[scene_name]()
    si = _copy_si(scene.si, size_x=width, size_y=height, **si_changes)
    eye = np.array([131.0, 77.0, -15000.0], f32)      # off-axis: no exactly zero direction component
    look = np.array([57.0, 23.0, 0.0], f32)
    w = f32(6400.0)
    ys, xs = np.mgrid[0:height, 0:width]
    ratio = f32(width) / f32(height)
    step_x = f32(ratio * w / f32(width))
    step_y = f32(w / f32(height))
    tx = (look[0] - step_x * (xs - width // 2).astype(f32)).astype(f32)
    ty = (look[1] + step_y * (ys - height // 2).astype(f32)).astype(f32)
    targets = np.stack([tx, ty, np.full_like(tx, look[2])], -1).reshape(-1, 3).astype(f32)
    n = len(targets)
    origins = np.repeat(eye[None, :], n, 0).astype(f32)
    index = (ys * width + xs).reshape(-1).astype(i32)
    return dict(name="launch", scene=scene, si=si, ppi=scene.ppi, origins=origins, targets=targets, index=index,
                width=width, height=height)


def case_post(seed=21, width=96, height=64, pp_type=0, iteration=0, param1=0.0, param2=0.0, param3=0, period=0,
              plateaus=0):
    """the reference's own post-processing kernels on a synthetic frame buffer with depth discontinuities.
    period: the random buffer repeats with that period - with 900, randoms[i + 100] == randoms[i + 1000], the one
    statement in which the two engines' k_depthOfField differ (CRT:1101 / CL:2968) cannot show, and the reference's
    image is the CUDA engine's on every pixel"""
    rng = np.random.default_rng(seed)
    solr = _solr()
    n = width * height
    pp = np.zeros(n, solr.PP_DTYPE)
    ys, xs = np.mgrid[0:height, 0:width]
    colour = np.stack([(xs / width), (ys / height), 0.5 + 0.5 * np.sin(xs * 0.3)], -1).reshape(-1, 3)
    scale = (iteration - 10 + 1) if iteration > 10 else 1
    pp["colorInfo"][:, :3] = (colour * scale * rng.uniform(0.7, 1.3, (n, 3))).astype(f32)
    depth = 8000.0 + 4000.0 * ((xs // 12 + ys // 9) % 3) + rng.uniform(-50, 50, (height, width))
    if plateaus:
        # plateaus of one depth, `plateaus` pixels wide: a pixel whose taps all land on its own plateau, or on farther
        # ones, is occluded by nothing - there the two engines' k_ambientOcclusion leave the colour alone and agree
        depth = 8000.0 + 4000.0 * ((xs // plateaus + ys // (3 * plateaus // 4)) % 3) + 0.0 * depth
    pp["colorInfo"][:, 3] = depth.reshape(-1).astype(f32)
    randoms = rng.uniform(-1.0, 1.0, max(n, 4096)).astype(f32)
    if period:
        randoms = np.ascontiguousarray(np.resize(randoms[:period], len(randoms)))
    si = _scene_info(size_x=width, size_y=height, pathTracingIteration=iteration)
    ppi = solr.PostProcessingInfo(pp_type, param1, param2, param3)
    return dict(name="post", si=si, ppi=ppi, pp=pp, randoms=randoms, width=width, height=height)


def _p(a):
    return C.c_void_p(a.ctypes.data)


# ---- evaluation on both sides ------------------------------------------------------------------------------
def reference_outputs(case, variant):
    """runs the probe kernel of `case` on the GPU; returns a dict of arrays laid out like oracle_outputs'"""
    co = PROBES.get(variant)   # the post-processing kernels come from the renderer's own code object
    name = case["name"]
    if name == "box":
        n = len(case["origins"])
        hit = np.zeros(n, i32)
        run_kernel(co, "probe_box", [("in", cl_boxes(case["boxes"])), ("in", f4(case["origins"])),
                                     ("in", f4(case["directions"])), ("in", case["t0"]), ("in", case["t1"]),
                                     ("out", hit)], [n])
        return dict(hit=hit)
    if name == "primitive":
        n = len(case["origins"])
        out = np.zeros((n, 4, 4), f32)
        initial = np.zeros((n, 2, 4), f32)
        initial[:, :, :3] = case["initial"]
        run_kernel(co, "probe_primitive", [val(case["si"]), ("in", cl_prims(case["prims"])), ("in", case["materials"]),
                                           ("in", case["textures"]), ("in", f4(case["origins"])),
                                           ("in", f4(case["directions"])), ("in", case["shadows"]), ("in", initial),
                                           ("out", out)], [n])
        return dict(hit=(out[:, 3, 0] != 0).astype(i32), intersection=out[:, 0, :3].copy(), normal=out[:, 1, :3].copy(),
                    areas=out[:, 2, :3].copy(), shadow=out[:, 3, 1].copy(), w=out[:, :3, 3].copy())
    if name in ("closest", "shadow", "shader", "launch"):
        s = case["scene"]
        scene_args = [("in", cl_boxes(s.boxes)), val(len(s.boxes)), ("in", cl_prims(s.prims)), val(len(s.prims))]
    if name == "closest":
        n = len(case["origins"])
        out = np.zeros((n, 4, 4), f32)
        ids = np.zeros((n, 2), i32)
        run_kernel(co, "probe_closest", [val(case["si"])] + scene_args +
                   [("in", s.materials), ("in", s.textures), ("in", f4(case["origins"])), ("in", f4(case["targets"])),
                    ("in", case["iteration"]), ("in", case["current"]), ("out", out), ("out", ids)], [n])
        return dict(hit=ids[:, 0].copy(), primitive=ids[:, 1].copy(), intersection=out[:, 0, :3].copy(),
                    normal=out[:, 1, :3].copy(), areas=out[:, 2, :3].copy())
    if name == "shadow":
        n = len(case["origins"])
        out = np.zeros((n, 4), f32)
        run_kernel(co, "probe_shadow", [val(case["si"])] + scene_args +
                   [("in", s.materials), ("in", s.textures), ("in", f4(case["lamps"])), ("in", f4(case["origins"])),
                    ("in", case["object_id"]), ("in", case["iteration"]), ("out", out)], [n])
        return dict(result=out[:, 0].copy(), color=out[:, 1:4].copy())
    if name == "shader":
        n = len(case["origins"])
        inout = np.zeros((n, 5, 4), f32)
        inout[:, 0, :3] = case["normal"]
        inout[:, 1, :3] = case["inter"]
        inout[:, 2, :3] = case["closest_color"]
        inout[:, 3, :3] = case["total_blinn"]
        inout[:, 4, :] = case["attributes"]
        out = np.zeros((n, 3, 4), f32)
        randoms = s.randoms if len(s.randoms) else np.zeros(16, f32)
        run_kernel(co, "probe_shader", [val(case["si"]), val(case["ppi"])] + scene_args +
                   [("in", cl_lights(s.lights)), val(len(s.lights)), val(s.nb_lamps), ("in", s.materials),
                    ("in", s.textures), ("in", randoms), ("in", case["index"]), ("in", f4(case["origins"])),
                    ("in", case["object_id"]), ("in", f4(case["areas"])), ("in", case["iteration"]),
                    ("inout", inout), ("out", out)], [n])
        return dict(returned=out[:, 0, :3].copy(), shadow=out[:, 2, 0].copy(), normal=inout[:, 0, :3].copy(),
                    closest_color=inout[:, 2, :3].copy(), total_blinn=inout[:, 3, :3].copy(),
                    attributes=inout[:, 4, :].copy())
    if name == "intersection_shader":
        n = len(case["inter"])
        attributes = case["attributes"].copy()
        out = np.zeros((n, 4, 4), f32)
        run_kernel(co, "probe_intersection_shader", [val(case["si"]), ("in", cl_prims(case["prims"])),
                                                     ("in", case["materials"]), ("in", case["textures"]),
                                                     ("in", f4(case["inter"])), ("in", f4(case["areas"])),
                                                     ("inout", attributes), ("out", out)], [n])
        return dict(color=out[:, 0, :].copy(), bump=out[:, 1, :3].copy(), specular=out[:, 2, :3].copy(),
                    advanced=out[:, 3, :1].copy(), attributes=attributes)
    if name == "skybox":
        n = len(case["origins"])
        out = np.zeros((n, 4), f32)
        run_kernel(co, "probe_skybox", [val(case["si"]), ("in", case["materials"]), ("in", case["textures"]),
                                        ("in", f4(case["origins"])), ("in", f4(case["targets"])), ("out", out)], [n])
        return dict(color=out[:, :3].copy())
    if name == "vectors":
        n = len(case["incident"])
        out = np.zeros((n, 2, 4), f32)
        run_kernel(co, "probe_vectors", [("in", f4(case["incident"])), ("in", f4(case["normals"])), ("in", case["n1"]),
                                         ("in", case["n2"]), ("out", out)], [n])
        return dict(refracted=out[:, 0, :3].copy(), reflected=out[:, 1, :3].copy())
    if name == "make_color":
        n = len(case["colors"])
        bitmap = np.zeros(n * 3, np.uint8)
        run_kernel(co, "probe_make_color", [val(case["si"]), ("in", f4(case["colors"])), ("out", bitmap)], [n])
        return dict(bitmap=bitmap)
    if name == "launch":
        n = len(case["origins"])
        out = np.zeros((n, 4), f32)
        ids = np.zeros((n, 4), i32)
        randoms = s.randoms if len(s.randoms) else np.zeros(16, f32)
        run_kernel(co, "probe_launch", scene_args +
                   [("in", cl_lights(s.lights)), val(len(s.lights)), val(s.nb_lamps), ("in", s.materials),
                    ("in", s.textures), ("in", randoms), val(case["si"]), val(case["ppi"]),
                    ("in", f4(case["origins"])), ("in", f4(case["targets"])), ("in", case["index"]), ("out", out),
                    ("inout", ids)], [n])
        return dict(color=out[:, :3].copy(), depth=out[:, 3].copy(), ids=ids)
    if name == "post":
        # the reference's kernels as they are, from the renderer's own code object
        w, h = case["width"], case["height"]
        bitmap = np.zeros(w * h * 3, np.uint8)
        occ = (C.c_int * 2)(1, 1)
        t = case["ppi"].type
        if t == 1:
            run_kernel(RENDERER, "k_depthOfField", [val(occ), val(case["si"]), val(case["ppi"]), ("in", case["pp"]),
                                                    ("in", case["randoms"]), ("out", bitmap)], [w, h], [8, 8])
        elif t == 2:
            run_kernel(RENDERER, "k_ambientOcclusion", [val(occ), val(case["si"]), val(case["ppi"]), ("in", case["pp"]),
                                                        ("in", case["randoms"]), ("out", bitmap)], [w, h], [8, 8])
        else:
            run_kernel(RENDERER, "k_default", [val(occ), val(case["si"]), ("in", case["pp"]), ("out", bitmap)], [w, h],
                       [8, 8])
        return dict(bitmap=bitmap)
    raise KeyError(name)


def oracle_outputs(case):
    """the same evaluations through the CPU oracle in its OpenCL dialect"""
    L = loader.lib()
    was = L.oracle_get_dialect()
    L.oracle_set_dialect(1)
    try:
        return _oracle_outputs(L, case)
    finally:
        L.oracle_set_dialect(was)


def _oracle_outputs(L, case):
    name = case["name"]
    if name == "box":
        n = len(case["origins"])
        hit = np.zeros(n, i32)
        L.oracle_probe_box(n, _p(case["boxes"]), _p(case["origins"]), _p(case["directions"]), _p(case["t0"]),
                           _p(case["t1"]), _p(hit))
        return dict(hit=hit)
    if name == "primitive":
        n = len(case["origins"])
        inter = np.ascontiguousarray(case["initial"][:, 0, :]).copy()
        normal = np.ascontiguousarray(case["initial"][:, 1, :]).copy()
        areas, shadow, hit = np.zeros((n, 3), f32), np.zeros(n, f32), np.zeros(n, i32)
        L.oracle_probe_primitive(n, C.byref(case["si"]), _p(case["prims"]), _p(case["materials"]), _p(case["textures"]),
                                 _p(case["origins"]), _p(case["directions"]), _p(case["shadows"]), _p(inter), _p(normal),
                                 _p(areas), _p(shadow), _p(hit))
        return dict(hit=hit, intersection=inter, normal=normal, areas=areas, shadow=shadow)
    if name in ("closest", "shadow", "shader", "launch"):
        s = case["scene"]
        osc = s.oracle_scene()
    if name == "closest":
        n = len(case["origins"])
        hit, prim = np.zeros(n, i32), np.zeros(n, i32)
        inter, normal, areas = np.zeros((n, 3), f32), np.zeros((n, 3), f32), np.zeros((n, 3), f32)
        L.oracle_probe_closest(n, C.byref(osc), C.byref(case["si"]), _p(case["origins"]), _p(case["targets"]),
                               _p(case["iteration"]), _p(case["current"]), _p(hit), _p(prim), _p(inter), _p(normal),
                               _p(areas))
        return dict(hit=hit, primitive=prim, intersection=inter, normal=normal, areas=areas)
    if name == "shadow":
        n = len(case["origins"])
        result, color = np.zeros(n, f32), np.zeros((n, 3), f32)
        # the OpenCL dialect leaves out the lamp only, whatever this argument says; the CUDA dialect also the primitive
        # the point lies on, when the case names it (`shaded`: the moot cases) - otherwise nobody, as the reference's probe
        shaded = case["shaded"] if "shaded" in case else np.full(n, -12345, i32)
        L.oracle_probe_shadow(n, C.byref(osc), C.byref(case["si"]), _p(case["lamps"]), _p(case["origins"]),
                              _p(case["object_id"]), _p(shaded), _p(case["iteration"]), _p(result), _p(color))
        return dict(result=result, color=color)
    if name == "shader":
        n = len(case["origins"])
        normal = case["normal"].copy()
        cc, tb, at = case["closest_color"].copy(), case["total_blinn"].copy(), case["attributes"].copy()
        ret, shadow = np.zeros((n, 3), f32), np.zeros(n, f32)
        status = L.oracle_probe_shader(n, C.byref(osc), C.byref(case["si"]), _p(case["index"]), _p(case["origins"]),
                                       _p(normal), _p(case["object_id"]), _p(case["inter"]), _p(case["areas"]), _p(cc),
                                       _p(case["iteration"]), _p(tb), _p(at), _p(ret), _p(shadow))
        assert status == 0, "the oracle read outside the random buffer"
        return dict(returned=ret, shadow=shadow, normal=normal, closest_color=cc, total_blinn=tb, attributes=at)
    if name == "intersection_shader":
        n = len(case["inter"])
        at = case["attributes"].copy()
        color, bump, spec, adv = np.zeros((n, 4), f32), np.zeros((n, 3), f32), np.zeros((n, 4), f32), np.zeros((n, 4), f32)
        L.oracle_probe_intersection_shader(n, C.byref(case["si"]), _p(case["prims"]), _p(case["materials"]),
                                           _p(case["textures"]), _p(case["inter"]), _p(case["areas"]), _p(at), _p(color),
                                           _p(bump), _p(spec), _p(adv))
        return dict(color=color, bump=bump, specular=spec[:, :3].copy(), advanced=adv[:, :1].copy(), attributes=at)
    if name == "skybox":
        n = len(case["origins"])
        color = np.zeros((n, 3), f32)
        L.oracle_probe_skybox(n, C.byref(case["si"]), _p(case["materials"]), _p(case["textures"]), _p(case["origins"]),
                              _p(case["targets"]), _p(color))
        return dict(color=color)
    if name == "vectors":
        n = len(case["incident"])
        refracted, reflected = np.zeros((n, 3), f32), np.zeros((n, 3), f32)
        L.oracle_probe_vectors(n, _p(case["incident"]), _p(case["normals"]), _p(case["n1"]), _p(case["n2"]),
                               _p(refracted), _p(reflected))
        return dict(refracted=refracted, reflected=reflected)
    if name == "make_color":
        n = len(case["colors"])
        bitmap = np.zeros(n * 3, np.uint8)
        L.oracle_probe_make_color(n, C.byref(case["si"]), _p(case["colors"]), _p(bitmap))
        return dict(bitmap=bitmap)
    if name == "launch":
        n = len(case["origins"])
        color, depth, ids = np.zeros((n, 3), f32), np.zeros(n, f32), np.zeros((n, 4), i32)
        status = L.oracle_probe_launch(n, C.byref(osc), C.byref(case["si"]), _p(case["origins"]), _p(case["targets"]),
                                       _p(case["index"]), _p(color), _p(depth), _p(ids))
        assert status == 0, "the oracle read outside the random buffer"
        return dict(color=color, depth=depth, ids=ids)
    if name == "post":
        w, h = case["width"], case["height"]
        bitmap = np.zeros(w * h * 3, np.uint8)
        ids = np.zeros((w * h, 4), i32)
        empty = np.zeros(16, np.uint8)
        osc = loader.OracleScene(None, 0, None, 0, None, 0, 0, None, empty.ctypes.data, case["randoms"].ctypes.data,
                                 len(case["randoms"]))
        status = L.oracle_postprocess(C.byref(osc), C.byref(case["si"]), C.byref(case["ppi"]), _p(case["pp"]), _p(ids),
                                      _p(bitmap))
        assert status == 0
        return dict(bitmap=bitmap)
    raise KeyError(name)


def both_dialects(case):
    """(CUDA dialect, OpenCL dialect) outputs of the oracle for a case"""
    L = loader.lib()
    assert L.oracle_get_dialect() == 0
    return _oracle_outputs(L, case), oracle_outputs(case)


def dialects_agree(case, per_key=False):
    """per element: do the oracle's two dialects return the same bits?  Where they do, none of the statements in which
    the reference's engines differ can be seen in the result, and the reference's own (OpenCL) output is what the CUDA
    engine computes as well - the elements on which the product can be held to reference output directly.
    per_key: a dict of masks, one per output (a pixel whose depth is under a switch may still have an unaffected
    colour); otherwise one mask over all outputs."""
    cuda, cl = both_dialects(case)
    masks = {}
    for key in cuda:
        same = same_bits(cuda[key], cl[key])
        masks[key] = same.reshape(len(same), -1).all(axis=1)
    if per_key:
        return masks
    return np.logical_and.reduce(list(masks.values()))


def _keep_where_the_dialects_agree(case, n):
    """the first n elements of a case on which the two dialects agree in every output (input selection only: both
    sides of a comparison are then handed the same arrays)"""
    agree = dialects_agree(case)
    m = len(agree)
    keep = np.flatnonzero(agree)[:n]
    assert len(keep) >= min(n, 256), "only %d of %d candidate elements are free of dialect switches" % (len(keep), m)
    out = {k: (v[keep].copy() if isinstance(v, np.ndarray) and len(v) == m and k not in ("materials", "textures") else v)
           for k, v in case.items()}
    out["candidates"] = m
    out["kept_of_candidates"] = int(agree.sum())
    return out


def _hash_array(h, a):
    """named fields only: the padding of the records (8 bytes in a box, 8 in a primitive) is never written"""
    a = np.ascontiguousarray(a)
    if a.dtype.names:
        for f in a.dtype.names:
            h.update(np.ascontiguousarray(a[f]).tobytes())
    else:
        h.update(a.tobytes())


def input_digest(case):
    """sha1 over the input arrays of a case: a fixture is only compared with the inputs it was made from"""
    h = hashlib.sha1()
    for key in sorted(case):
        v = case[key]
        if isinstance(v, np.ndarray):
            h.update(key.encode())
            _hash_array(h, v)
        elif isinstance(v, C.Structure):
            h.update(key.encode())
            h.update(bytes(v))
        elif isinstance(v, SceneData):
            for a in (v.boxes, v.prims, v.lights, v.textures, v.materials[:4096]):
                _hash_array(h, a)
    return h.hexdigest()


# ---- what "agree" means -------------------------------------------------------------------------------------
def same_bits(a, b):
    """equal as numbers (+0 == -0), NaN only equal to NaN"""
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return (a == b) | (np.isnan(a) & np.isnan(b))
    return a == b


def close(a, b, rel, floor=1.0):
    """per element (row) of a and b: the largest component difference <= rel * the largest component magnitude
    (at least `floor`): a vector is judged against its own length, not component by component"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    a2, b2 = a.reshape(len(a), -1), b.reshape(len(b), -1)
    scale = np.maximum(np.maximum(np.abs(a2).max(axis=1), np.abs(b2).max(axis=1)), floor)
    diff = np.abs(a2 - b2)
    diff[np.isnan(a2) & np.isnan(b2)] = 0.0
    return diff.max(axis=1) <= rel * scale


# ---- the cases the fixtures hold and the tests run ---------------------------------------------------------
CASES = {
    "box": lambda: case_box(),
    "primitive": lambda: case_primitive(),
    "primitive_double_sided": lambda: case_primitive(double_sided=1),
    "primitive_all_triangles": lambda: case_primitive(extended=0),
    "primitive_colour_key": lambda: case_primitive(transparent_color=0.7),
    "closest_cornell": lambda: case_closest("cornell"),
    "closest_mix": lambda: case_closest("mix"),
    "closest_sticks": lambda: case_closest("sticks"),
    "closest_triangles": lambda: case_closest("triangles"),
    "closest_triangles_flat": lambda: case_closest("triangles_flat"),
    "closest_textured": lambda: case_closest("textured"),
    "shadow_cornell": lambda: case_shadow("cornell"),
    "shadow_mix": lambda: case_shadow("mix"),
    "shadow_sticks": lambda: case_shadow("sticks"),
    "shadow_cornell_opaque_moot": lambda: case_shadow("cornell_opaque", moot=True),
    "shadow_sticks_moot": lambda: case_shadow("sticks", moot=True),
    "shadow_triangles_moot": lambda: case_shadow("triangles", moot=True),
    "shader_cornell": lambda: case_shader("cornell"),
    "shader_cornell_blinn_only": lambda: case_shader("cornell", graphicsLevel=2),
    "shader_cornell_accumulation": lambda: case_shader("cornell", pathTracingIteration=12, timestamp=17),
    "shader_mix": lambda: case_shader("mix"),
    "shader_textured": lambda: case_shader("textured"),
    "shader_mix_one_lamp": lambda: case_shader("mix_one_lamp"),
    "shader_cornell_opaque_moot": lambda: case_shader("cornell_opaque", moot=True),
    "shader_sticks_moot": lambda: case_shader("sticks", moot=True),
    "shader_triangles_moot": lambda: case_shader("triangles", moot=True, pathTracingIteration=3),
    "intersection_shader": lambda: case_intersection_shader(),
    "skybox": lambda: case_skybox(),
    "vectors": lambda: case_vectors(),
    "make_color_rgb": lambda: case_make_color(frame_buffer_type=0),
    "make_color_bgr": lambda: case_make_color(frame_buffer_type=1),
    "launch_cornell": lambda: case_launch("cornell"),
    "launch_cornell_fog": lambda: case_launch("cornell", atmosphericEffect=1, viewDistance=36000.0),
    "launch_mix": lambda: case_launch("mix"),
    "launch_textured": lambda: case_launch("textured"),
    "launch_cornell_opaque": lambda: case_launch("cornell_opaque", nbRayIterations=1),
    "launch_sticks": lambda: case_launch("sticks"),
    "launch_mesh_100": lambda: case_launch("mesh_100"),
    "launch_sticks_backdrop": lambda: case_launch("sticks_backdrop"),
    "launch_mix_one_lamp": lambda: case_launch("mix_one_lamp"),
    "post_default": lambda: case_post(pp_type=0),
    "post_default_accumulated": lambda: case_post(pp_type=0, iteration=13),
    "post_depth_of_field": lambda: case_post(pp_type=1, param1=9000.0, param2=300.0, param3=16),
    "post_depth_of_field_moot": lambda: case_post(pp_type=1, param1=9000.0, param2=300.0, param3=16, period=900),
    "post_ambient_occlusion": lambda: case_post(pp_type=2, param2=40.0),
    "post_ambient_occlusion_plateaus": lambda: case_post(pp_type=2, param2=10.0, width=256, height=192, plateaus=64),
}
