"""The mesh kernel launch by launch, one frame at a time, after 600 frames with three in flight: the order of the launch is\nre-made every 64th frame (sixteenth, when this was recorded) from the costs the tiles reported, and with it the set of tiles rendered as four quadrant waves -\nthe kernel takes 0.40 or 0.46 ms by turns (and 0.61 for the first eight, before the first order for one frame in flight).\nusage: python tools/mesh_kernel_phases.py  -> profiles/r6/mesh_kernel_per_launch.txt"""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
W, H = 1920, 1080
k = solr.Kernel(engine="hip", deterministic_seed=1)
solr.scenes.height_field(k, width=W, height=H)
L = k.L
image = np.zeros((H, W, 3), np.uint8)
L.SolR_RunKernel(0.0, image.ctypes.data)
def frames(n, flights):
    L.SolRx_SetFramesInFlight(flights)
    for _ in range(n):
        L.SolRx_Render(0.0)
    L.SolRx_FlushFrames()
frames(600, 3)
L.SolRx_SetFramesInFlight(1)
hip.solr_hip_kernel_time(None, 1)
hip.solr_hip_enable_timing(1)
splits = []
for i in range(80):
    L.SolR_RunKernel(0.0, image.ctypes.data)
    splits.append(hip.solr_hip_split_tiles())
hip.solr_hip_enable_timing(0)
samples = (C.c_float * 128)()
hip.solr_hip_timing_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
got = hip.solr_hip_timing_samples(samples, None, 128)
print("per-launch ms after switching from 3 frames in flight to 1:", " ".join("%.3f" % samples[i] for i in range(got)))
print("tiles rendered as four quadrant waves, by the order of that launch:", " ".join(str(v) for v in splits))
k.finalize()
