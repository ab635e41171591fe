"""Development aid: where a wave's cycles go (node loop / leaves / the rest of either walk / shading and the rest).

Needs the timing build of the engine library (shader-clock reads around the regions; they perturb what they
measure - every read drains the wave's outstanding scalar loads - so read the split, not the total):
    make -C sol-r_amd -B EXTRA_HIPFLAGS=-DSOLR_TIMING csrc/libsolr_hip.so && python tools/wave_time_split.py --scene cornell
"""
import argparse, ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--iterations", type=int, default=None)
ap.add_argument("--frames", type=int, default=8)
a = ap.parse_args()
hip = solr.hip_lib()
if not hasattr(hip, "solr_hip_wave_cycles"):
    sys.exit("not the timing build: make -C sol-r_amd -B EXTRA_HIPFLAGS=-DSOLR_TIMING csrc/libsolr_hip.so")
k = solr.Kernel(engine="hip")
kw = dict(width=a.width, height=a.height)
if a.iterations is not None:
    kw["iterations"] = a.iterations
if hasattr(solr.scenes, a.scene):
    getattr(solr.scenes, a.scene)(k, **kw)
else:           # the test scenes: textured, primitives_mix, triangles_only, sticks ...
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import scenes_extra
    getattr(scenes_extra, a.scene)(k, **kw)
hip.solr_hip_set_frames_in_flight(1)
for _ in range(4):
    k.render()
out = (C.c_ulonglong * 16)()
hip.solr_hip_wave_cycles(out, 1)
import time
hip.solr_hip_synchronize()
t0 = time.perf_counter()
for _ in range(a.frames):
    k.render()
hip.solr_hip_synchronize()
ms = (time.perf_counter() - t0) * 1e3 / a.frames
hip.solr_hip_wave_cycles(out, 1)
total, closest, shadow, node, leaf, nadv, nleaf, waves, shade, trace, epilogue = [float(v) for v in out][:11]
k.finalize()
per = lambda v: v / waves  # noqa: E731
print("scene %s %dx%d: %d waves per frame, %.3f ms per frame in this build; shader-clock cycles per wave" % (a.scene, a.width, a.height, waves / a.frames, ms))
print("  whole kernel           %9.0f" % per(total))
print("  closest-hit walks      %9.0f  %5.1f %%" % (per(closest), 100 * closest / total))
print("  shadow walks           %9.0f  %5.1f %%" % (per(shadow), 100 * shadow / total))
print("    node loop            %9.0f  %5.1f %%   %6.1f calls per wave, %6.0f cycles per call" % (per(node), 100 * node / total, per(nadv), node / max(nadv, 1)))
print("    leaves               %9.0f  %5.1f %%   %6.1f visits per wave, %6.0f cycles per visit" % (per(leaf), 100 * leaf / total, per(nleaf), leaf / max(nleaf, 1)))
print("    walk set-up          %9.0f  %5.1f %%" % (per(closest + shadow - node - leaf), 100 * (closest + shadow - node - leaf) / total))
print("  shading, camera, output %8.0f  %5.1f %%" % (per(total - closest - shadow), 100 * (total - closest - shadow) / total))
print("    primitiveShader without its shadow walks %8.0f  %5.1f %%" % (per(shade - shadow), 100 * (shade - shadow) / total))
print("    the rest of launchRayTracing (bounce bookkeeping, sky, blend) %8.0f  %5.1f %%" % (per(trace - closest - shade), 100 * (trace - closest - shade) / total))
print("    camera set-up (kernel start to the trace) %8.0f  %5.1f %%" % (per(total - trace - epilogue), 100 * (total - trace - epilogue) / total))
print("    stores and the rest (end of the trace to the end) %8.0f  %5.1f %%" % (per(epilogue), 100 * epilogue / total))
