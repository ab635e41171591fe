"""Development aid: the reference's OpenCL k_standardRenderer (oracle/_ref, compiled for gfx950 as it is)
timed on the same GPU and frame as the HIP engine."""
import os, sys, importlib, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
solr = importlib.import_module("sol-r_amd")
from oracle import loader
hip = solr.hip_lib()
for scene, kw in (("cornell", dict(width=1920, height=1080, iterations=3)), ("height_field", dict(width=1920, height=1080)),
                  ("molecule", dict(width=1920, height=1080))):
    if len(sys.argv) > 1 and scene not in sys.argv[1:]:
        continue
    k = solr.Kernel(engine="hip", deterministic_seed=1)
    getattr(solr.scenes, scene)(k, **kw)
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    t = {}
    loader.ref_render(flat, si, ppi, eye, direction, angles, repeats=6, timing=t)
    k.render()
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    hip.solr_hip_synchronize(); hip.solr_hip_kernel_time(None, 1); hip.solr_hip_enable_timing(1)
    for _ in range(10):
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
    n = C.c_int(0); ms = hip.solr_hip_kernel_time(C.byref(n), 1) / 10; hip.solr_hip_enable_timing(0)
    print("%s 1920x1080: reference OpenCL k_standardRenderer %.3f ms, HIP engine %.3f ms, ratio %.1fx" % (scene, t["renderer_ms"], ms, t["renderer_ms"] / ms))
    k.finalize()
