"""Development aid: per-tile (= per-wave) start/end times of one frame.
Prints the wave-duration distribution, the utilisation timeline and where the slow tiles are."""
import sys, os, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
solr = importlib.import_module("sol-r_amd")
scenes = importlib.import_module("sol-r_amd.scenes")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--iterations", type=int, default=None)
ap.add_argument("--tile-scheduling", type=int, default=1)
ap.add_argument("--strip", type=int, nargs=2, default=None, help="rank world")
a = ap.parse_args()
k = solr.Kernel("hip", deterministic_seed=1)
kw = dict(width=a.width, height=a.height)
if a.iterations:
    kw["iterations"] = a.iterations
getattr(scenes, a.scene)(k, **kw)
hip = solr.hip_lib()
H_FULL = a.height
if a.strip:
    first, count, per = solr.strip_rows(a.strip[0], a.strip[1], a.height)
    hip.solr_hip_set_strip(first, count)
    a.height = count
hip.solr_hip_set_tile_scheduling(a.tile_scheduling)
import ctypes as C
k.render()
flat = k.flat_scene(); si, ppi, eye, direction, angles = k.frame_parameters(); si.pathTracingIteration = 0
objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
fp = lambda v: v.ctypes.data_as(C.POINTER(C.c_float))
def frame():
    hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
for _ in range(12):
    frame()
    hip.solr_hip_synchronize()
hip.solr_hip_enable_tile_clocks(1)
frame()
hip.solr_hip_synchronize()
print("scene %s %dx%d, tile scheduling mode %d, cost-ordered launch active: %d" % (a.scene, a.width, a.height, a.tile_scheduling, hip.solr_hip_tile_scheduling_active()))
tx, ty = (a.width + 7) // 8, (a.height + 7) // 8
clk = np.zeros((tx * ty, 2), dtype=np.uint64)
n = hip.solr_hip_tile_clocks(clk.ctypes.data, tx * ty)
hip.solr_hip_enable_tile_clocks(0)
clk = clk[:n].astype(np.int64)
t0 = clk[:, 0].min()
start = (clk[:, 0] - t0) / 100.0  # microseconds
end = (clk[:, 1] - t0) / 100.0
dur = end - start
total = end.max()
print("tiles %d  frame %.1f us  sum(wave time) %.0f us  mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f us" % (
    n, total, dur.sum(), dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
edges = np.linspace(0, total, 21)
for i in range(20):
    lo, hi = edges[i], edges[i + 1]
    ov = np.clip(np.minimum(end, hi) - np.maximum(start, lo), 0, None).sum() / (hi - lo)
    print("  %6.0f-%6.0f us: %6.0f waves in flight, %5d started" % (lo, hi, ov, ((start >= lo) & (start < hi)).sum()))
d2 = dur.reshape(ty, tx)
print("mean wave time by tile-row band (top to bottom):", np.round(d2.reshape(ty, tx).mean(axis=1)[::max(ty // 15, 1)], 1))
last = np.argsort(end)[-10:]
print("last tiles to finish (tx, ty, start, dur):", [(int(i % tx), int(i // tx), round(float(start[i])), round(float(dur[i]))) for i in last])
