"""Soak of the image streaming: thousands of 1080p frames under a camera that moves, every frame's image as it arrived in bands
(SolRx_Render into m_bitmap / SolR_RunKernel into the caller's array, in turn) against the device's image read back behind the
kernel (solr_hip_d2h).  usage: python tools/stream_soak.py [scene] [frames]  -> profiles/r6/stream_soak.txt"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
W, H = 1920, 1080
hip = solr.hip_lib()
k = solr.Kernel(engine="hip", deterministic_seed=1)
kw = dict(width=W, height=H)
if scene == "cornell":
    kw["iterations"] = 3
getattr(solr.scenes, scene)(k, **kw)
L = k.L
caller = np.zeros((H, W, 3), np.uint8)
plain, ids = np.zeros((H, W, 3), np.uint8), np.zeros((H, W, 4), np.int32)
L.SolR_RunKernel(0.0, caller.ctypes.data)
before = hip.solr_hip_stream_next_image(-2)
wrong_frames = wrong_bytes = 0
distinct = set()
t0 = time.perf_counter()
for i in range(frames):
    k.set_camera((13.0 * (i % 400) - 2600.0, 7.0 * (i % 173), -15000.0 + 11.0 * (i % 97)))
    if i % 2:
        assert L.SolR_RunKernel(0.0, caller.ctypes.data) == 0
        got = caller
    else:
        assert L.SolRx_Render(0.0) == 0
        ptr = L.SolRx_GetBitmap()
        got = np.frombuffer((C.c_ubyte * (W * H * 3)).from_address(ptr), np.uint8).reshape(H, W, 3)
    si = k.frame_parameters()[0]
    hip.solr_hip_d2h(C.byref(si), C.c_void_p(plain.ctypes.data), C.c_void_p(ids.ctypes.data))
    if not np.array_equal(got, plain):
        wrong_frames += 1
        wrong_bytes += int((got != plain).sum())
    if i % 97 == 0:
        distinct.add(int(plain[::16, ::16].astype(np.uint32).sum()))
k.check(0, "the soak")
print("%s 1920x1080: %d frames in %.1f s, %d of them left in bands; frames whose image differs from the device's: %d (%d bytes); "
      "%d of %d sampled frames distinct" % (scene, frames, time.perf_counter() - t0, hip.solr_hip_stream_next_image(-2) - before,
                                            wrong_frames, wrong_bytes, len(distinct), (frames + 96) // 97))
k.finalize()
sys.exit(1 if wrong_frames else 0)
