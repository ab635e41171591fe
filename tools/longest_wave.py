"""Development aid: where a frame's longest waves are and what they do (leaf entries per lane and walks per workgroup of
the walk replay, solr_hip_walk_bound with SOLR_HIP_WALK_BOUND_DUMP; tile durations from the engine's tile clocks).
    python tools/longest_wave.py [scene]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
scene = sys.argv[1] if len(sys.argv) > 1 else "height_field"
W, H = 1920, 1080
k = solr.Kernel(engine="hip")
kw = dict(width=W, height=H)
if scene == "cornell":
    kw["iterations"] = 3
getattr(solr.scenes, scene)(k, **kw)
hip.solr_hip_set_tile_scheduling(0)            # raster order: workgroup b is tile b
for _ in range(4):
    k.render()
flat = k.flat_scene()
si, ppi, eye, direction, angles = k.frame_parameters()
si.pathTracingIteration = 0
objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
hip.solr_hip_enable_tile_clocks(1)
hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
hip.solr_hip_synchronize()
tiles = ((W + 7) // 8) * ((H + 7) // 8)
clocks = np.zeros((tiles, 2), np.uint64)
got = hip.solr_hip_tile_clocks(C.c_void_p(clocks.ctypes.data), tiles)
hip.solr_hip_enable_tile_clocks(0)
dur = (clocks[:got, 1] - clocks[:got, 0]).astype(np.float64) / 100.0      # 100 MHz ticks -> microseconds
path = "/tmp/walk_bound_dump.bin"
os.environ["SOLR_HIP_WALK_BOUND_DUMP"] = path
ms, stats = (C.c_double * 3)(), (C.c_ulonglong * 4)()
status = hip.solr_hip_walk_bound(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles), 3, ms, stats)
k.check(status, "solr_hip_walk_bound")
raw = open(path, "rb").read()
n = int(np.frombuffer(raw[:4], np.uint32)[0])
visits = np.frombuffer(raw[4:4 + 4 * 64 * n], np.uint32).reshape(n, 64)
heads = np.frombuffer(raw[4 + 4 * 64 * n:4 + 4 * 64 * n + 16 * n], np.int32).reshape(n, 4)
order = np.argsort(-dur)[:8]
print("%s: %d tiles, frame's tile durations: median %.1f us, 99th percentile %.1f us, longest %.1f us" % (
    scene, got, np.median(dur), np.percentile(dur, 99), dur.max()))
print("leaf entries per lane over all workgroups: median %d, 99th percentile %d, max %d" % (
    np.median(visits.max(axis=1)), np.percentile(visits.max(axis=1), 99), visits.max()))
for t in order:
    tx, ty = t % ((W + 7) // 8), t // ((W + 7) // 8)
    print("  tile (%d, %d): %.1f us, %d walks, leaf entries per lane min / median / max %d / %d / %d" % (
        tx, ty, dur[t], heads[t, 0], visits[t].min(), np.median(visits[t]), visits[t].max()))
k.finalize()
