"""sol-r_amd - MI355X (gfx950) engine for Sol-R's per-pixel rendering path.

This Python package is plumbing only: it builds and loads the two native
libraries and exposes their C entry points through ctypes so that tests,
``bench.py`` and ``__graft_entry__.py`` can drive them.

* ``csrc/libsolr_hip.so``  - hand-written HIP kernels behind the C-ABI of
  ``include/solr_hip.h`` (the reference's ``CudaRayTracer.h`` boundary).
* ``host/libsolr.so``      - C++ host mirror of the reference's ``GPUKernel`` /
  ``SolRStub`` interface (scene store, box-tree builder, frame protocol).

There is no CPU rendering path here: without the HIP library and a GPU every
render call fails loudly (``SolrError``).  The CPU oracle lives in ``oracle/``
and is never imported from this package.

The directory name contains a hyphen, so import it with::

    import importlib; solr = importlib.import_module("sol-r_amd")
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HIP_LIB = os.environ.get("SOLR_HIP_LIB") or os.path.join(_HERE, "csrc", "libsolr_hip.so")
HOST_LIB = os.environ.get("SOLR_HOST_LIB") or os.path.join(_HERE, "host", "libsolr.so")  # sanitizer builds

# ---- constants (include/solr_types.h) ---------------------------------------
NB_MAX_ITERATIONS = 10
NB_MAX_MATERIALS = 65506 + 30
MAX_BITMAP_SIZE = 1920 * 1080
MATERIAL_NONE = -1
TEXTURE_NONE = -1
TEXTURE_MANDELBROT = -2
TEXTURE_JULIA = -3

ptSphere, ptCylinder, ptTriangle, ptCheckboard, ptCamera, ptXYPlane, ptYZPlane, ptXZPlane = range(8)
ptMagicCarpet, ptEnvironment, ptEllipsoid, ptQuad, ptCone = 8, 9, 10, 11, 12
ctPerspective, ctOrthographic, ctAnaglyph, ctVR, ctPanoramic, ctAntialiazed, ctVolumeRendering = range(7)
glNoShading, glPhong, glPhongAndBlinn, glReflectionsAndRefractions, glFull = range(5)
aiNone, aiBasic, aiFull, aiRandomIllumination = range(4)
aeNone, aeFog = 0, 1
ftRGB, ftBGR = 0, 1
ppe_none, ppe_depthOfField, ppe_ambientOcclusion, ppe_radiosity, ppe_filter, ppe_cartoon = range(6)


class SolrError(RuntimeError):
    """Raised when the native engine reports an error (or is missing)."""


# ---- ctypes mirrors of the by-pointer records --------------------------------
class SceneInfo(C.Structure):
    _fields_ = [
        ("size_x", C.c_int), ("size_y", C.c_int),
        ("cameraType", C.c_int), ("graphicsLevel", C.c_int),
        ("nbRayIterations", C.c_int), ("transparentColor", C.c_float),
        ("viewDistance", C.c_float), ("shadowIntensity", C.c_float),
        ("eyeSeparation", C.c_float), ("renderBoxes", C.c_int),
        ("pathTracingIteration", C.c_int), ("maxPathTracingIterations", C.c_int),
        ("frameBufferType", C.c_int), ("timestamp", C.c_int),
        ("atmosphericEffect", C.c_int), ("doubleSidedTriangles", C.c_int),
        ("extendedGeometry", C.c_int), ("advancedIllumination", C.c_int),
        ("draftMode", C.c_int), ("skyboxRadius", C.c_int),
        ("skyboxMaterialId", C.c_int), ("gradientBackground", C.c_int),
        ("geometryEpsilon", C.c_float), ("rayEpsilon", C.c_float),
        ("backgroundColor", C.c_float * 4),
    ]


class PostProcessingInfo(C.Structure):
    _fields_ = [("type", C.c_int), ("param1", C.c_float), ("param2", C.c_float), ("param3", C.c_int)]


class Vec4i(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("z", C.c_int), ("w", C.c_int)]


assert C.sizeof(SceneInfo) == 112 and C.sizeof(PostProcessingInfo) == 16

# ---- numpy views of the flattened arrays --------------------------------------
f4, i4 = np.float32, np.int32
BOX_DTYPE = np.dtype({"names": ["min", "max", "nbPrimitives", "startIndex", "indexForNextBox"],
                      "formats": [(f4, 3), (f4, 3), i4, i4, (i4, 2)],
                      "offsets": [0, 12, 24, 28, 32], "itemsize": 48})
PRIMITIVE_DTYPE = np.dtype({"names": ["p0", "p1", "p2", "n0", "n1", "n2", "size", "type", "index", "materialId",
                                      "vt0", "vt1", "vt2"],
                            "formats": [(f4, 3)] * 7 + [i4, i4, i4] + [(f4, 2)] * 3,
                            "offsets": [0, 12, 24, 36, 48, 60, 72, 84, 88, 92, 96, 104, 112], "itemsize": 128})
MATERIAL_DTYPE = np.dtype({"names": ["innerIllumination", "color", "specular", "reflection", "refraction",
                                     "transparency", "opacity", "attributes", "textureMapping", "textureOffset",
                                     "textureIds", "advancedTextureOffset", "advancedTextureIds", "mappingOffset"],
                           "formats": [(f4, 4)] * 3 + [f4] * 4 + [(i4, 4)] * 6 + [(f4, 2)],
                           "offsets": [0, 16, 32, 48, 52, 56, 60, 64, 80, 96, 112, 128, 144, 160], "itemsize": 176})
LIGHT_DTYPE = np.dtype({"names": ["primitiveId", "materialId", "location", "color"],
                        "formats": [i4, i4, (f4, 3), (f4, 4)], "offsets": [0, 4, 8, 32], "itemsize": 48})
PP_DTYPE = np.dtype({"names": ["colorInfo", "sceneInfo"], "formats": [(f4, 4), (f4, 4)], "offsets": [0, 16],
                     "itemsize": 32})


def build(verbose=False):
    """Compile both native libraries for gfx950 (hipcc cross-compiles without a GPU)."""
    # (the renderer's instantiations are one object per row of the launch table: built side by side)
    jobs = str(max(1, min(8, len(os.sched_getaffinity(0)))))
    res = subprocess.run(["make", "-C", _HERE, "-j", jobs, "all"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise SolrError("native build failed")
    return HIP_LIB, HOST_LIB


_hip = None
_host = None


def hip_lib():
    """The C-ABI boundary library (include/solr_hip.h)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise SolrError("%s is missing: run __graft_entry__.build() (there is no fallback engine)" % HIP_LIB)
        _hip = C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL)
        _declare_hip(_hip)
    return _hip


def host_lib():
    """The host mirror of the reference interface (SolR_* and SolRx_*)."""
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(HOST_LIB):
            raise SolrError("%s is missing: run __graft_entry__.build()" % HOST_LIB)
        _host = C.CDLL(HOST_LIB)
        _declare_host(_host)
    return _host


def _declare_hip(L):
    P = C.POINTER
    L.solr_hip_last_error.argtypes = [C.c_char_p, C.c_int]
    L.solr_hip_last_error.restype = C.c_int
    L.solr_hip_clear_error.argtypes = []
    L.solr_hip_device_count.restype = C.c_int
    L.solr_hip_set_gpu_count.argtypes = [C.c_int]
    L.solr_hip_gpu_count.restype = C.c_int
    L.solr_hip_set_device.argtypes = [C.c_int]
    L.solr_hip_set_stream.argtypes = [C.c_void_p]
    L.solr_hip_synchronize.argtypes = []
    L.solr_hip_set_strip.argtypes = [C.c_int, C.c_int]
    for name in ("solr_hip_device_bitmap", "solr_hip_device_primitive_ids", "solr_hip_device_postprocessing"):
        getattr(L, name).restype = C.c_void_p
    L.solr_hip_bind_device_bitmap.argtypes = [C.c_void_p]
    L.solr_hip_d2h_postprocessing.argtypes = [C.c_void_p]
    L.solr_hip_h2d_postprocessing.argtypes = [C.c_void_p, C.c_void_p]
    L.solr_hip_initialize.argtypes = [P(SceneInfo)]
    L.solr_hip_reshape.argtypes = [P(SceneInfo)]
    L.solr_hip_render.argtypes = [P(SceneInfo), P(Vec4i), P(PostProcessingInfo), P(C.c_float), P(C.c_float),
                                  P(C.c_float)]
    L.solr_hip_render_counting.argtypes = L.solr_hip_render.argtypes + [P(C.c_ulonglong)]
    L.solr_hip_walk_bound.argtypes = L.solr_hip_render.argtypes + [C.c_int, P(C.c_double), P(C.c_ulonglong)]
    L.solr_hip_walk_bound.restype = C.c_int
    L.solr_hip_walk_bound_lists.argtypes = [P(C.c_ulonglong)]
    L.solr_hip_walk_bound_lists.restype = None
    L.solr_hip_walk_records_keep.argtypes = [C.c_int]
    L.solr_hip_walk_records_keep.restype = None
    L.solr_hip_walk_records_info.argtypes = [P(C.c_ulonglong)]
    L.solr_hip_walk_records_info.restype = C.c_int
    L.solr_hip_walk_records_copy.argtypes = [C.c_void_p, C.c_uint, C.c_int]
    L.solr_hip_walk_records_copy.restype = C.c_int
    L.solr_hip_walk_replay.argtypes = [C.c_uint, C.c_long, C.c_int, P(C.c_double), P(C.c_ulonglong)]
    L.solr_hip_walk_replay.restype = C.c_int
    L.solr_hip_walk_records_release.argtypes = []
    L.solr_hip_walk_records_release.restype = None
    L.solr_hip_set_short_ray_lists.argtypes = [C.c_int]
    L.solr_hip_set_short_ray_lists.restype = None
    L.solr_hip_short_ray_lists.restype = C.c_int
    L.solr_hip_d2h.argtypes = [P(SceneInfo), C.c_void_p, C.c_void_p]
    L.solr_hip_enable_timing.argtypes = [C.c_int]
    L.solr_hip_kernel_time.argtypes = [P(C.c_int), C.c_int]
    L.solr_hip_kernel_time.restype = C.c_double
    L.solr_hip_timing_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.solr_hip_set_frames_in_flight.argtypes = [C.c_int]
    L.solr_hip_get_frames_in_flight.restype = C.c_int
    L.solr_hip_flight_stream.argtypes = [C.c_int]
    L.solr_hip_flight_stream.restype = C.c_void_p
    L.solr_hip_next_flight.restype = C.c_int
    L.solr_hip_set_flight_streams.argtypes = [P(C.c_void_p), C.c_int]
    L.solr_hip_set_tile_scheduling.argtypes = [C.c_int]
    L.solr_hip_tile_scheduling_active.restype = C.c_int
    L.solr_hip_enable_tile_clocks.argtypes = [C.c_int]
    L.solr_hip_tile_clocks.argtypes = [C.c_void_p, C.c_int]
    L.solr_hip_tile_clocks.restype = C.c_int
    L.solr_hip_set_variant.argtypes = [C.c_int]
    L.solr_hip_get_strip.argtypes = [P(C.c_int), P(C.c_int)]
    L.solr_hip_comm_unique_id.argtypes = [C.c_void_p]
    L.solr_hip_comm_init.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.solr_hip_comm_ranks.restype = C.c_int
    L.solr_hip_comm_shared_seed.restype = C.c_uint
    L.solr_hip_gather_strips.argtypes = [C.c_int]
    L.solr_hip_gather_ids.argtypes = [C.c_int]
    L.solr_hip_gathered_frame.restype = C.c_void_p
    L.solr_hip_d2h_gathered.argtypes = [C.c_void_p]
    L.solr_hip_d2h_gathered_ids.argtypes = [C.c_void_p]
    L.solr_hip_h2d_randoms_sized.argtypes = [C.c_void_p, C.c_long]
    L.solr_hip_d2h_image_async.restype = C.c_int
    L.solr_hip_stream_next_image.argtypes = [C.c_int]
    L.solr_hip_stream_next_image.restype = C.c_int
    L.solr_hip_d2h_streamed_image.argtypes = [C.c_void_p]
    L.solr_hip_d2h_streamed_image.restype = C.c_int
    L.solr_hip_d2h_streamed.argtypes = [C.c_void_p, C.c_void_p]
    L.solr_hip_d2h_streamed.restype = C.c_int
    L.solr_hip_image_share.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.solr_hip_image_share.restype = C.c_int
    L.solr_hip_d2h_gathered_async.restype = C.c_int
    L.solr_hip_image_wait.argtypes = [C.c_int]
    L.solr_hip_image_wait.restype = C.c_void_p
    L.solr_hip_get_variant.restype = C.c_int
    L.solr_hip_memory_usage.argtypes = [P(C.c_ulonglong)]
    L.solr_hip_set_movable.argtypes = [C.c_void_p, C.c_int]
    L.solr_hip_rotate_primitives.argtypes = [P(C.c_float), P(C.c_float), P(C.c_float), C.c_float]
    L.solr_hip_rotate_primitives.restype = C.c_int
    L.solr_hip_device_rotations.restype = C.c_int
    L.solr_hip_read_nodes.argtypes = [C.c_int, C.c_void_p, C.c_int]
    L.solr_hip_read_nodes.restype = C.c_int
    L.solr_hip_read_primitives.argtypes = [C.c_void_p, C.c_int]
    L.solr_hip_read_primitives.restype = C.c_int
    # the by-value reference entry points are exercised from C++ (host/HipKernel.cpp);
    # ctypes cannot 16-byte align a by-value struct, so they get no argtypes here
    L.h2d_scene.argtypes = [C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.h2d_materials.argtypes = [C.c_uint64, C.c_void_p, C.c_int]
    L.h2d_randoms.argtypes = [C.c_uint64, C.c_void_p]
    L.h2d_lightInformation.argtypes = [C.c_uint64, C.c_void_p, C.c_int]
    L.finalize_scene.argtypes = [C.c_uint64]


def _declare_host(L):
    d, i = C.c_double, C.c_int
    P = C.POINTER
    L.SolR_SetSceneInfo.argtypes = [i, i, i, i, d, d, d, d, d, d, d, d, i, i, i, i, i, i, i, i, i, i, i, i, d, d]
    L.SolR_SetPostProcessingInfo.argtypes = [i, d, d, i]
    L.SolR_SetDraftMode.argtypes = [i]
    L.SolR_InitializeKernel.argtypes = [C.c_bool, i, i]
    L.SolR_SetCamera.argtypes = [d] * 9
    L.SolR_SetCamera.restype = None
    L.SolRx_SetCameraW.argtypes = [d] * 10
    L.SolRx_SetCameraW.restype = None
    L.SolR_RunKernel.argtypes = [d, C.c_void_p]
    L.SolR_AddPrimitive.argtypes = [i, i]
    L.SolR_SetPrimitive.argtypes = [i] + [d] * 12 + [i]
    L.SolR_GetPrimitive.argtypes = [i] + [P(d)] * 12 + [P(i)]
    L.SolR_GetPrimitiveAt.argtypes = [i, i]
    L.SolR_GetPrimitiveCenter.argtypes = [i, P(d), P(d), P(d)]
    L.SolR_RotatePrimitives.argtypes = [i, i] + [d] * 6
    L.SolR_SetPrimitiveMaterial.argtypes = [i, i]
    L.SolR_GetPrimitiveMaterial.argtypes = [i]
    L.SolR_SetPrimitiveNormals.argtypes = [i] + [d] * 9
    L.SolR_SetPrimitiveTextureCoordinates.argtypes = [i] + [d] * 6
    L.SolR_SetMaterial.argtypes = [i, d, d, d, d, d, d, i, i, i, d, d, i, i, i, i, i, i, i, d, d, d, d, d, d, i]
    L.SolR_CompactBoxes.argtypes = [C.c_bool]
    L.SolR_GetLight.argtypes = [i]
    L.SolR_SetTexture.argtypes = [i, C.c_void_p, i, i, i, i]
    L.SolR_GetTextureSize.argtypes = [i, P(i), P(i), P(i)]
    L.SolR_GetNbTextures.argtypes = [P(i)]
    L.SolRx_SelectEngine.argtypes = [C.c_char_p]
    L.SolRx_SetDeterministic.argtypes = [C.c_long]
    L.SolRx_LastError.argtypes = [C.c_char_p, i]
    L.SolRx_Render.argtypes = [d]
    L.SolRx_SetFramesInFlight.argtypes = [i]
    L.SolRx_SetGpuCount.argtypes = [i]
    L.SolRx_SetGpuCount.restype = i
    L.SolRx_GetBitmap.restype = C.c_void_p
    for name in ("SolRx_GetBoxes", "SolRx_GetPrimitives", "SolRx_GetMaterials", "SolRx_GetRandoms",
                 "SolRx_GetPrimitiveIds"):
        getattr(L, name).argtypes = [P(C.c_void_p), P(i)]
    L.SolRx_GetLights.argtypes = [P(C.c_void_p), P(i), P(i)]
    L.SolRx_GetTextureAtlas.argtypes = [P(C.c_void_p), P(C.c_long)]
    L.SolRx_GetSceneInfo.argtypes = [P(SceneInfo), P(PostProcessingInfo), P(C.c_float), P(C.c_float), P(C.c_float)]
    L.SolRx_GetPostProcessingBuffer.argtypes = [C.c_void_p]
    L.SolRx_AddRectangle.argtypes = [d] * 6 + [i]
    L.SolRx_SetSceneInfoExtras.argtypes = [i, i]
    L.SolRx_GetMovable.argtypes = [P(C.c_void_p), P(i)]
    L.SolR_LoadFromFile.argtypes = [C.c_char_p, d]
    L.SolR_GetMaterial.argtypes = [i] + [P(d)] * 6 + [P(i)] * 3 + [P(d)] * 2 + [P(i)] * 7 + [P(d)] * 6 + [P(i)]
    L.SolR_GetTexture.argtypes = [i, C.c_void_p]
    L.SolR_RotatePrimitive.argtypes = [i] + [d] * 6
    L.SolR_RecompileKernels.argtypes = [C.c_char_p]
    L.SolR_SaveToFile.argtypes = [C.c_char_p]
    L.SolR_LoadMolecule.argtypes = [C.c_char_p, i, d, d, i, d]
    L.SolR_LoadOBJModel.argtypes = [C.c_char_p, i, i, d, i, P(d)]
    L.SolRx_LoadSWCMorphology.argtypes = [C.c_char_p] + [d] * 7 + [i]


def _np_from_ptr(ptr, count, dtype):
    if not ptr or count <= 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


class FlatScene:
    """Flattened arrays exactly as the host hands them to the device layer."""

    def __init__(self, boxes, primitives, lights, nb_lamps, materials, randoms, textures):
        self.boxes, self.primitives, self.lights = boxes, primitives, lights
        self.nb_lamps, self.materials, self.randoms, self.textures = nb_lamps, materials, randoms, textures


SCENE_DEFAULTS = dict(
    width=512, height=512, graphicsLevel=glFull, nbRayIterations=3, transparentColor=2.0, viewDistance=50000.0,
    shadowIntensity=1.0, eyeSeparation=380.0, bgColor=(0.0, 0.0, 0.0, 0.5), renderBoxes=0, pathTracingIteration=0,
    maxPathTracingIterations=100, frameBufferType=ftRGB, timestamp=0, atmosphericEffect=aeNone,
    cameraType=ctPerspective, doubleSidedTriangles=0, extendedGeometry=1, advancedIllumination=aiNone,
    skyboxSize=45000, skyboxMaterialId=MATERIAL_NONE, geometryEpsilon=0.001, rayEpsilon=0.05,
    gradientBackground=0, draftMode=0)
"""Defaults of the reference's Scene::initialize (apps/scenes/Scene.cpp:199-228) except that the skybox is off
and transparentColor lets axis planes be hit (the viewer's 0 makes every plane fully transparent,
GeometryIntersections.cuh:561)."""


class Kernel:
    """Thin object wrapper over the flat SolR_* API (one singleton engine per process)."""

    def __init__(self, engine="hip", deterministic_seed=12345, device=0):
        self.L = host_lib()
        self.L.SolRx_SelectEngine(engine.encode())
        self.engine = engine
        self.seed = deterministic_seed
        self.device = device
        self.info = dict(SCENE_DEFAULTS)
        self.pp = dict(type=ppe_none, param1=0.0, param2=0.0, param3=0)
        self.initialized = False

    # -- errors ---------------------------------------------------------------
    def check(self, status=0, what="engine"):
        buf = C.create_string_buffer(512)
        code = self.L.SolRx_LastError(buf, 512)
        if status != 0 or code != 0:
            raise SolrError("%s failed (status %s, code %s): %s" % (what, status, code, buf.value.decode()))

    # -- scene info -------------------------------------------------------------
    def set_scene_info(self, **kw):
        unknown = set(kw) - set(self.info)
        if unknown:
            raise KeyError("unknown SceneInfo fields: %s" % sorted(unknown))
        self.info.update(kw)
        s = self.info
        bg = s["bgColor"]
        self.L.SolR_SetSceneInfo(
            s["width"], s["height"], s["graphicsLevel"], s["nbRayIterations"], s["transparentColor"],
            s["viewDistance"], s["shadowIntensity"], s["eyeSeparation"], bg[0], bg[1], bg[2], bg[3],
            s["renderBoxes"], s["pathTracingIteration"], s["maxPathTracingIterations"], s["frameBufferType"],
            s["timestamp"], s["atmosphericEffect"], s["cameraType"], s["doubleSidedTriangles"],
            s["extendedGeometry"], s["advancedIllumination"], s["skyboxSize"], s["skyboxMaterialId"],
            s["geometryEpsilon"], s["rayEpsilon"])
        self.L.SolRx_SetSceneInfoExtras(s["gradientBackground"], s["draftMode"])

    def set_post_processing(self, type=ppe_none, param1=0.0, param2=0.0, param3=0):
        self.pp = dict(type=type, param1=param1, param2=param2, param3=param3)
        self.L.SolR_SetPostProcessingInfo(type, param1, param2, param3)

    def initialize(self, device=None, **scene_info):
        device = self.device if device is None else device
        self.set_scene_info(**scene_info)
        self.L.SolR_SetPostProcessingInfo(self.pp["type"], self.pp["param1"], self.pp["param2"], self.pp["param3"])
        if self.engine == "hip":
            hip_lib().solr_hip_clear_error()
            hip_lib().solr_hip_set_device(device)
        status = self.L.SolR_InitializeKernel(False, 0, device)
        self.check(status, "SolR_InitializeKernel")
        self.L.SolRx_SetDeterministic(self.seed)
        self.initialized = True

    def finalize(self):
        self.L.SolR_FinalizeKernel()
        self.initialized = False

    def set_gpu_count(self, n):
        """the frame shared out over n devices of this process (occupancyParameters.x); returns the number in use"""
        got = self.L.SolRx_SetGpuCount(n)
        self.check(0 if got >= 1 else -1, "SolRx_SetGpuCount")
        return got

    # -- materials / primitives ---------------------------------------------------
    def add_material(self, r=1.0, g=1.0, b=1.0, noise=0.0, reflection=0.0, refraction=0.0, procedural=False,
                     wireframe=False, wireframeWidth=0, transparency=0.0, opacity=0.0, diffuseTextureId=TEXTURE_NONE,
                     normalTextureId=TEXTURE_NONE, bumpTextureId=TEXTURE_NONE, specularTextureId=TEXTURE_NONE,
                     reflectionTextureId=TEXTURE_NONE, transparencyTextureId=TEXTURE_NONE,
                     ambientOcclusionTextureId=TEXTURE_NONE, specValue=0.1, specPower=200.0, specCoef=0.0,
                     innerIllumination=0.0, illuminationDiffusion=None, illuminationPropagation=None,
                     fastTransparency=False):
        """SolR_AddMaterial + SolR_SetMaterial.  Illumination defaults follow createRandomMaterials
        (apps/scenes/Scene.cpp:381-382): diffusion = 10 * viewDistance, propagation = viewDistance."""
        vd = self.info["viewDistance"]
        if illuminationDiffusion is None:
            illuminationDiffusion = vd * 10
        if illuminationPropagation is None:
            illuminationPropagation = vd
        idx = self.L.SolR_AddMaterial()
        self.L.SolR_SetMaterial(idx, r, g, b, noise, reflection, refraction, int(procedural), int(wireframe),
                                wireframeWidth, transparency, opacity, diffuseTextureId, normalTextureId,
                                bumpTextureId, specularTextureId, reflectionTextureId, transparencyTextureId,
                                ambientOcclusionTextureId, specValue, specPower, specCoef, innerIllumination,
                                illuminationDiffusion, illuminationPropagation, int(fastTransparency))
        return idx

    def add_primitive(self, ptype, p0, p1=(0, 0, 0), p2=(0, 0, 0), size=(0, 0, 0), material=0, movable=1):
        idx = self.L.SolR_AddPrimitive(ptype, movable)
        self.L.SolR_SetPrimitive(idx, p0[0], p0[1], p0[2], p1[0], p1[1], p1[2], p2[0], p2[1], p2[2], size[0],
                                 size[1], size[2], material)
        return idx

    def set_normals(self, idx, n0, n1, n2):
        self.L.SolR_SetPrimitiveNormals(idx, *n0, *n1, *n2)

    def set_texture_coordinates(self, idx, t0, t1, t2):
        self.L.SolR_SetPrimitiveTextureCoordinates(idx, *t0, *t1, *t2)

    def set_texture(self, index, pixels, texture_type=0):
        a = np.ascontiguousarray(pixels, dtype=np.uint8)
        h, w, d = a.shape
        if self.L.SolR_SetTexture(index, a.ctypes.data, w, h, d, texture_type) != 0:
            raise SolrError("SolR_SetTexture failed")

    def compact_boxes(self, reconstruct=True):
        return self.L.SolR_CompactBoxes(reconstruct)

    def load_from_file(self, path, scale):
        """Append an .irt scene dump (reference: FileMarshaller::loadFromFile via SolR_LoadFromFile): its
        primitives, textures and materials; every primitive of the kernel is then rescaled so that the
        loaded model is `scale` high."""
        return self.L.SolR_LoadFromFile(os.fsencode(path), scale)

    def load_obj_model(self, path, material_id=0, auto_scale=True, scale=5000.0, auto_center=True):
        """Append a Wavefront OBJ model and its MTL materials (reference: OBJReader::loadModelFromFile via
        SolR_LoadOBJModel).  Returns -(scaled height) / 2, the ground level the reference's scenes use."""
        height = C.c_double()
        self.L.SolR_LoadOBJModel(os.fsencode(path), material_id, 1 if auto_scale else 0, scale,
                                 1 if auto_center else 0, C.byref(height))
        return height.value

    def load_molecule(self, path, geometry_type=0, atom_size=100.0, stick_size=10.0, material_type=0, scale=100.0):
        """Append a molecule from a PDB file (reference: PDBReader::loadAtomsFromFile via SolR_LoadMolecule);
        overwrites materials 0..118 with the element colours."""
        return self.L.SolR_LoadMolecule(os.fsencode(path), geometry_type, atom_size, stick_size, material_type, scale)

    def load_swc_morphology(self, path, position=(0.0, 0.0, 0.0), scale=(1.0, 1.0, 1.0, 1.0), material_id=0):
        """Append a neuron morphology (reference: SWCReader::loadMorphologyFromFile): spheres and
        cylinders along the sample points.  Returns the number of samples read."""
        return self.L.SolRx_LoadSWCMorphology(os.fsencode(path), *position, *scale, material_id)

    def save_to_file(self, path):
        return self.L.SolR_SaveToFile(os.fsencode(path))

    def rotate_primitives(self, center=(0.0, 0.0, 0.0), angles=(0.0, 0.0, 0.0)):
        """One step of an animated scene: GPUKernel::rotatePrimitives + compactBoxes(false), as the
        reference's scenes do per frame (SolR_RotatePrimitives).  With the HIP engine and an unchanged
        scene the rotation runs on the resident scene and the host copy follows lazily."""
        return self.L.SolR_RotatePrimitives(0, 0, center[0], center[1], center[2], angles[0], angles[1], angles[2])

    def pending_rotations(self):
        return self.L.SolRx_PendingRotations()

    def sync_host(self):
        self.L.SolRx_SyncHost()

    def device_nodes(self, exact=True, order_free=None):
        """The resident node records (n, 2, 4) float32 as the device holds them now: the reference's
        node list (exact), the engine's walk-order list, or its order-free list of octant `order_free`
        (0 ... 7; empty when there are none)."""
        hip = hip_lib()
        which = (2 + order_free) if order_free is not None else (1 if exact else 0)
        cap = hip.solr_hip_read_nodes(which, None, 0)
        buf = np.zeros((max(cap, 1), 4), np.float32)
        if cap < 0 or hip.solr_hip_read_nodes(which, buf.ctypes.data, cap) != cap:
            raise SolrError("solr_hip_read_nodes failed")
        return buf[:cap].reshape(-1, 2, 4)

    def device_primitives(self):
        """The resident primitive records (n, 8, 4) float32."""
        hip = hip_lib()
        cap = hip.solr_hip_read_primitives(None, 0)
        buf = np.zeros((max(cap, 1), 4), np.float32)
        if cap < 0 or hip.solr_hip_read_primitives(buf.ctypes.data, cap) != cap:
            raise SolrError("solr_hip_read_primitives failed")
        return buf[:cap].reshape(-1, 8, 4)

    def set_camera(self, eye, look_at=(0, 0, 0), angles=(0, 0, 0), w=6400.0):
        self.L.SolRx_SetCameraW(eye[0], eye[1], eye[2], look_at[0], look_at[1], look_at[2], angles[0], angles[1],
                                angles[2], w)

    # -- rendering ------------------------------------------------------------------
    def render(self, **scene_info):
        """One frame through render_begin / render_end; returns the RGB8 image (H, W, 3)."""
        if scene_info:
            self.set_scene_info(**scene_info)
        w, h = self.info["width"], self.info["height"]
        image = np.zeros((h, w, 3), dtype=np.uint8)
        status = self.L.SolR_RunKernel(0.0, image.ctypes.data)
        self.check(status, "SolR_RunKernel")
        return image

    def postprocessing_buffer(self):
        """Float framebuffer of the last frame: (H, W, 8) = colorInfo.xyzw, sceneInfo.xyzw."""
        w, h = self.info["width"], self.info["height"]
        out = np.zeros((h, w, 8), dtype=np.float32)
        status = self.L.SolRx_GetPostProcessingBuffer(out.ctypes.data)
        self.check(status, "SolRx_GetPostProcessingBuffer")
        return out

    def primitive_ids(self):
        ptr, n = C.c_void_p(), C.c_int()
        self.L.SolRx_GetPrimitiveIds(C.byref(ptr), C.byref(n))
        w, h = self.info["width"], self.info["height"]
        return _np_from_ptr(ptr.value, n.value * 4, np.dtype(np.int32)).reshape(h, w, 4)

    def primitive_at(self, x, y):
        return self.L.SolR_GetPrimitiveAt(x, y)

    # -- flattened arrays (inputs of the device layer; what the oracle is given) ---------
    def flat_scene(self):
        L = self.L
        ptr, n, n2 = C.c_void_p(), C.c_int(), C.c_int()
        L.SolRx_GetBoxes(C.byref(ptr), C.byref(n))
        boxes = _np_from_ptr(ptr.value, n.value, BOX_DTYPE)
        L.SolRx_GetPrimitives(C.byref(ptr), C.byref(n))
        prims = _np_from_ptr(ptr.value, n.value, PRIMITIVE_DTYPE)
        L.SolRx_GetLights(C.byref(ptr), C.byref(n), C.byref(n2))
        lights = _np_from_ptr(ptr.value, n.value, LIGHT_DTYPE)
        nb_lamps = n2.value
        L.SolRx_GetMaterials(C.byref(ptr), C.byref(n))
        materials = _np_from_ptr(ptr.value, n.value, MATERIAL_DTYPE)
        L.SolRx_GetRandoms(C.byref(ptr), C.byref(n))
        randoms = _np_from_ptr(ptr.value, n.value, np.dtype(np.float32))
        nbytes = C.c_long()
        L.SolRx_GetTextureAtlas(C.byref(ptr), C.byref(nbytes))
        textures = _np_from_ptr(ptr.value, nbytes.value, np.dtype(np.uint8))
        return FlatScene(boxes, prims, lights, nb_lamps, materials, randoms, textures)

    def frame_parameters(self):
        """(SceneInfo, PostProcessingInfo, eye[3], dir[3], angles[4]) of the next frame."""
        si, pp = SceneInfo(), PostProcessingInfo()
        eye, d, ang = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 4)()
        self.L.SolRx_GetSceneInfo(C.byref(si), C.byref(pp), eye, d, ang)
        return si, pp, np.array(eye, dtype=np.float32), np.array(d, dtype=np.float32), np.array(ang, np.float32)


# ---- multi-GPU: framebuffer row strips + one gather ------------------
def strip_rows(rank, world, height):
    """Rows [first, first + count) of the image that rank `rank` of `world` renders, and the
    common strip height used as the gather's slot size.  The reference splits the frame into
    contiguous row strips per device (CudaRayTracer.cu:1694-1696, d2h_bitmap :1650-1670); unlike
    it, the last strip absorbs a remainder instead of dropping rows when height % world != 0."""
    rows_per_rank = (height + world - 1) // world
    first = rank * rows_per_rank
    count = max(0, min(rows_per_rank, height - first))
    return first, count, rows_per_rank


def strip_row_costs(height):
    """What each row of this process's strip cost in the frame rendered last (float32[height], 0 outside the
    strip): solr_hip_strip_row_costs."""
    import numpy as np
    cost = np.zeros(height, np.float32)
    if hip_lib().solr_hip_strip_row_costs(cost.ctypes.data_as(C.POINTER(C.c_float)), height) != 0:
        raise SolrError("solr_hip_strip_row_costs failed")
    return cost


def balanced_strips(row_cost, world, align=8):
    """[(first_row, nb_rows)] * world: contiguous strips of equal cost (solr_hip_balanced_strips; pure arithmetic,
    no GPU needed).  row_cost: the cost of every row of the frame, summed over the ranks."""
    import numpy as np
    cost = np.ascontiguousarray(row_cost, np.float32)
    first, count = (C.c_int * world)(), (C.c_int * world)()
    if hip_lib().solr_hip_balanced_strips(cost.ctypes.data_as(C.POINTER(C.c_float)), len(cost), world, align,
                                          first, count) != 0:
        raise ValueError("solr_hip_balanced_strips: bad arguments")
    return list(zip(first, count))


def set_strip_table(strips, height):
    """the strips of all ranks ([(first_row, nb_rows)] in rank order) for the library's gather and halo exchange;
    None forgets the table (back to the equal strips of strip_rows)"""
    if strips is None:
        return hip_lib().solr_hip_set_strip_table(None, None, 0, 0)
    world = len(strips)
    first, count = (C.c_int * world)(*[s[0] for s in strips]), (C.c_int * world)(*[s[1] for s in strips])
    return hip_lib().solr_hip_set_strip_table(first, count, world, height)


def gather_strips(dist, torch, strip, rows_per_rank, width, height, rank, world, dst=0, slots=None, assemble=True,
                  strips=None):
    """One collective: every rank's RGB8 strip (a flat uint8 tensor of count*width*3 bytes) to `dst`.
    Returns the assembled (height, width, 3) image on `dst`, None elsewhere.  Works for the nccl
    (= RCCL over xGMI) and gloo backends.  `slots` lets the caller keep the receive buffers across
    frames; assemble=False skips the final concatenation (the strips then stay in `slots`).
    `strips` = [(first_row, nb_rows)] in rank order when they are not the equal ones of strip_rows (balanced_strips):
    the slots are then as large as the largest strip and the image is put together from their used parts."""
    if strips is not None:
        rows_per_rank = max(1, max(count for _, count in strips))
    slot = rows_per_rank * width * 3
    send = strip
    if strip.numel() != slot:  # the last strip may be shorter: gather needs equal slots
        send = torch.zeros((slot,), dtype=torch.uint8, device=strip.device)
        send[: strip.numel()].copy_(strip)
    if rank == dst and slots is None:
        slots = [torch.empty((slot,), dtype=torch.uint8, device=strip.device) for _ in range(world)]
    dist.gather(send, slots if rank == dst else None, dst=dst)
    if rank != dst or not assemble:
        return None
    if strips is not None:
        return torch.cat([slots[r][: count * width * 3] for r, (_, count) in enumerate(strips)]).reshape(height, width, 3)
    return torch.cat(slots)[: height * width * 3].reshape(height, width, 3)


class StripGather:
    """Pipelined assembly of the frame from the ranks' row strips.

    ONE collective per frame - a gather of the RGB8 strips to `dst` (RCCL over xGMI with the nccl
    backend, gloo on CPU) - issued asynchronously, so that it overlaps the rendering of the next
    frame.  `depth` strip buffers alternate; buffer(i) hands out the buffer of frame i only after
    the gather that last read it has completed (for nccl that wait is stream-ordered, the host
    does not block).  On `dst` the receive slots are consecutive views of one (rows, width, 3)
    tensor, so the assembled image needs no further copy.  Strips are padded to the common slot
    size (strip_rows): the engine writes its rows at the start of the buffer.

    pipelined=False issues the gather with blocking semantics on one buffer instead.  With the nccl
    backend of this PyTorch that runs the collective in order on the CURRENT stream - no event hop
    between streams - which on MI355X / ROCm 7 costs about 6 us of launch gaps per frame against two
    hops of about 16 us each for the pipelined form (profiles/r1/dist_overhead.txt): the better
    choice whenever a strip renders in less time than the hops cost, i.e. for the frames bench.py times."""

    def __init__(self, dist, torch, width, height, rank, world, device="cpu", dst=0, depth=2, pipelined=True):
        self.dist, self.torch = dist, torch
        self.pipelined = pipelined
        self.width, self.height, self.rank, self.world, self.dst, self.depth = width, height, rank, world, dst, depth
        self.first_row, self.nb_rows, self.rows_per_rank = strip_rows(rank, world, height)
        self.slot = self.rows_per_rank * width * 3
        self.strips = [torch.zeros((self.slot,), dtype=torch.uint8, device=device) for _ in range(depth)]
        self.frames = self.lists = None
        if rank == dst:
            self.frames = [torch.zeros((world * self.slot,), dtype=torch.uint8, device=device) for _ in range(depth)]
            self.lists = [[f[r * self.slot:(r + 1) * self.slot] for r in range(world)] for f in self.frames]
        self.works = [None] * depth

    def _wait(self, b):
        if self.works[b] is not None:
            self.works[b].wait()
            self.works[b] = None

    def buffer(self, i):
        """strip buffer of frame i (flat uint8, slot bytes); safe to overwrite when this returns"""
        b = i % self.depth
        self._wait(b)
        return self.strips[b]

    def submit(self, i):
        """start the gather of frame i's strip; returns immediately"""
        b = i % self.depth
        gather_list = self.lists[b] if self.rank == self.dst else None
        if not self.pipelined:
            self.dist.gather(self.strips[b], gather_list, dst=self.dst)
            return
        self.works[b] = self.dist.gather(self.strips[b], gather_list, dst=self.dst, async_op=True)

    def image(self, i):
        """the assembled (height, width, 3) image of frame i on `dst` (a view), None elsewhere"""
        b = i % self.depth
        self._wait(b)
        if self.rank != self.dst:
            return None
        return self.frames[b][: self.height * self.width * 3].reshape(self.height, self.width, 3)

    def drain(self):
        for b in range(self.depth):
            self._wait(b)


class StripPipeline:
    """One rank of the N-GPU frame loop: render this rank's strip, gather the strips on `dst`.

    frames_in_flight = K > 1: the engine renders consecutive frames on K streams of torch's pool
    (solr_hip_set_flight_streams; torch has spread its pool over the hardware queues), each with a strip
    buffer of its own, and the gather of a frame is issued IN ORDER ON THE STREAM THAT RENDERED IT, right
    behind the kernel: no event between streams anywhere, a strip buffer is reused only by the next frame
    of the same stream, and while stream f waits for its gather the other streams render.  A strip of a
    1080p frame is one round of waves whose tail leaves most of the chip idle:
    overlapping consecutive frames is what fills it.  Collectives are still issued in frame order, the
    same on every rank.  The host never blocks except to stay at most eight frames ahead.
    frames_in_flight = 1: one stream, render and gather in order on it."""

    def __init__(self, dist, torch, hip, width, height, rank, world, local_rank=0, dst=0, frames_in_flight=3):
        self.dist, self.torch, self.hip = dist, torch, hip
        self.flights = max(1, min(int(frames_in_flight), 4))
        self.device = local_rank
        self.streams = [torch.cuda.Stream(device=local_rank) for _ in range(self.flights)]
        torch.cuda.set_stream(self.streams[0])
        if self.flights > 1:
            pointers = (C.c_void_p * self.flights)(*[s.cuda_stream for s in self.streams])
            hip.solr_hip_set_flight_streams(pointers, self.flights)
            hip.solr_hip_set_frames_in_flight(self.flights)
        else:
            hip.solr_hip_set_frames_in_flight(1)
            hip.solr_hip_set_stream(C.c_void_p(self.streams[0].cuda_stream))
        self.sg = StripGather(dist, torch, width, height, rank, world, device="cuda:%d" % local_rank, dst=dst,
                              depth=self.flights, pipelined=False)
        hip.solr_hip_set_strip(self.sg.first_row, self.sg.nb_rows)
        hip.solr_hip_bind_device_bitmap(C.c_void_p(self.sg.strips[0].data_ptr()))
        self.throttle = [None, None]
        self.slot = {}
        self.frame_no = 0

    def frame(self, render):
        """render() issues one solr_hip_render / cudaRender of a first-pass frame; returns the frame number"""
        i = self.frame_no
        self.frame_no += 1
        torch = self.torch
        f = self.hip.solr_hip_next_flight() if self.flights > 1 else 0   # the stream / buffer set of this frame
        stream = self.streams[f]
        if i % 4 == 0:   # the host stays at most eight frames ahead of the GPU
            j = (i // 4) % 2
            if self.throttle[j] is not None:
                self.throttle[j].synchronize()
            self.throttle[j] = torch.cuda.Event()
            self.throttle[j].record(stream)
        self.hip.solr_hip_bind_device_bitmap(C.c_void_p(self.sg.strips[f].data_ptr()))
        render()
        torch.cuda.set_stream(stream)
        self.sg.submit(f)            # blocking-semantics gather: in order on `stream`, behind the kernel
        self.slot[i % 64] = f
        return i

    def image(self, i):
        """assembled frame i on dst (a view, valid until `flights` more frames are gathered); synchronises"""
        self.torch.cuda.synchronize(self.device)
        return self.sg.image(self.slot[i % 64])

    def drain(self):
        self.sg.drain()
        self.torch.cuda.synchronize(self.device)


from . import scenes  # noqa: E402,F401
