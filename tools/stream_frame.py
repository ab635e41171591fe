"""One frame at a time through the C ABI: cudaRender + d2h_bitmap (the reference's protocol) against cudaRender with the
image leaving in bands (solr_hip_stream_next_image + solr_hip_d2h_streamed_image); kernel time in both.
usage: python tools/stream_frame.py [scene] [frames]"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W, H = 1920, 1080
hip = solr.hip_lib()
k = solr.Kernel(engine="hip", deterministic_seed=1)
kw = dict(width=W, height=H)
if scene == "cornell":
    kw["iterations"] = 3
getattr(solr.scenes, scene)(k, **kw)
L = k.L
image = np.zeros((H, W, 3), np.uint8)
for _ in range(20):
    L.SolR_RunKernel(0.0, image.ctypes.data)


def block(call, n=frames, blocks=5):
    out = []
    for _ in range(blocks):
        hip.solr_hip_synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        out.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(out)


def kernel_ms(call, n=32):
    hip.solr_hip_kernel_time(None, 1)
    hip.solr_hip_enable_timing(1)
    for _ in range(n):
        call()
    hip.solr_hip_synchronize()
    hip.solr_hip_enable_timing(0)
    launches = C.c_int(0)
    hip.solr_hip_kernel_time.restype = C.c_double
    total = hip.solr_hip_kernel_time(C.byref(launches), 1)
    return total / max(launches.value, 1)


if os.environ.get("TILE_SCHEDULING"):
    hip.solr_hip_set_tile_scheduling(int(os.environ["TILE_SCHEDULING"]))
before = hip.solr_hip_stream_next_image(-2)
for name, call in (("SolR_RunKernel (the image into the caller's array)", lambda: L.SolR_RunKernel(0.0, image.ctypes.data)),
                   ("SolRx_Render (render_begin + render_end, the image into m_bitmap)", lambda: L.SolRx_Render(0.0))):
    t = block(call)
    print("%-70s %.4f ms per frame (median of 5 blocks of %d; %.4f ... %.4f); the kernel %.4f ms (HIP events)" % (
        name, t[2], frames, t[0], t[-1], kernel_ms(call)))
print("images that left in bands: %d; cost-ordered launch active: %d" % (hip.solr_hip_stream_next_image(-2) - before, hip.solr_hip_tile_scheduling_active()))
k.finalize()
