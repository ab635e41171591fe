"""Frame time through the reference's own frame protocol (render_begin / render_end via SolRx_Render,
and SolR_RunKernel which also copies the image to the caller), one frame at a time, as a viewer calls it.
usage: python tools/api_frame.py [--scene cornell] [--frames 200]"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell")
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
hip = solr.hip_lib()
k = solr.Kernel(engine="hip", deterministic_seed=1)
kw = dict(width=a.width, height=a.height)
if a.scene == "cornell":
    kw["iterations"] = 3
getattr(solr.scenes, a.scene)(k, **kw)
image = np.zeros((a.height, a.width, 3), np.uint8)
BLOCKS = 5      # blocks of a.frames frames; the median block is reported (the host thread is what is measured here)


def report(name, times):
    times = sorted(times)
    print("%-44s %.3f ms per frame (%s %dx%d, median of %d blocks of %d frames: %.3f ... %.3f)" % (
        name, 1e3 * times[len(times) // 2] / a.frames, a.scene, a.width, a.height, len(times), a.frames,
        1e3 * times[0] / a.frames, 1e3 * times[-1] / a.frames))


for _ in range(5):
    k.L.SolR_RunKernel(0.0, image.ctypes.data)
for name, call in (("SolRx_Render (render_begin + render_end)", lambda: k.L.SolRx_Render(0.0)),
                   ("SolR_RunKernel (+ image to the caller)", lambda: k.L.SolR_RunKernel(0.0, image.ctypes.data))):
    times = []
    for _ in range(BLOCKS):
        hip.solr_hip_synchronize()
        t0 = time.perf_counter()
        for _ in range(a.frames):
            call()
        times.append(time.perf_counter() - t0)
    report(name, times)
    print("    (images that have left in bands while their kernel rendered so far: %d)" % hip.solr_hip_stream_next_image(-2))
for flights in (2, 3):
    k.L.SolRx_SetFramesInFlight(flights)
    for name, call in (("SolRx_Render, %d frames in flight" % flights, lambda: k.L.SolRx_Render(0.0)),
                       ("SolR_RunKernel, %d frames in flight" % flights, lambda: k.L.SolR_RunKernel(0.0, image.ctypes.data))):
        times = []
        for _ in range(BLOCKS):
            for _ in range(8):
                call()
            t0 = time.perf_counter()
            for _ in range(a.frames):
                call()
            k.L.SolRx_FlushFrames()
            times.append(time.perf_counter() - t0)
        report(name, times)
k.L.SolRx_SetFramesInFlight(1)
t0 = time.perf_counter()
k.primitive_at(10, 10)
print("first getPrimitiveAt after a frame (ids read-back) %.3f ms" % (1e3 * (time.perf_counter() - t0)))
k.finalize()
