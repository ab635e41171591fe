"""Development aid: which path of k_ambientOcclusion do the pixels of a 3840 x 2160 frame take (debug build:
-DSOLR_AO_DEBUG writes the path into the image)?   SOLR_HIP_LIB=ab/libsolr_hip_aodebug.so python tools/ao_paths.py"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
solr = importlib.import_module("sol-r_amd")
import engine_probes as E
from oracle import probes
W, H = 3840, 2160
rng = np.random.default_rng(1)
pp = np.zeros(W * H, solr.PP_DTYPE)
pp["colorInfo"][:, :3] = 0.5
pp["colorInfo"][:, 3] = rng.uniform(5000, 20000, W * H).astype(np.float32)
randoms = (0.000005 * (rng.integers(0, 2000, 1920 * 1080) - 1000)).astype(np.float32)
si = probes._scene_info(size_x=W, size_y=H, pathTracingIteration=0)
ppi = solr.PostProcessingInfo(2, 11000.0, 10.0, 0)
case = dict(name="post", si=si, ppi=ppi, pp=pp, randoms=randoms, width=W, height=H)
if os.environ.get("AO_STATIC") == "1":
    solr.hip_lib().solr_hip_set_variant(9)
out = E.engine_outputs(solr, case)["bitmap"].reshape(H, W, 3)
STATIC = os.environ.get("AO_STATIC") == "1"    # solr_hip_set_variant(9): a fixed stride of tiles per workgroup, with the workgroups' clocks
# the workgroups' own clocks sit in the first three pixels of their first tiles: take them out before the statistics
# the workgroups' own clocks sit in the first four pixels of their first tiles (the fourth is a mark): take them out
# before the statistics
marks = []
for ty in range(H // 8):
    for tx in range(W // 32):
        yy, xx = ty * 8, tx * 32
        if tuple(int(v) for v in out[yy, xx + 3]) == (0xab, 0xcd, 0xef):
            marks.append([int(out[yy, xx + k, 0]) | int(out[yy, xx + k, 1]) << 8 | int(out[yy, xx + k, 2]) << 16 for k in range(3)])
            out[yy, xx:xx + 4] = out[yy, xx + 4]
marks = np.array(marks, dtype=np.int64)
t_begin = (marks[:, 0] - marks[:, 0].min()) % (1 << 24) * 0.01
t_end = (marks[:, 1] - marks[:, 0].min()) % (1 << 24) * 0.01
print("%d workgroups began their tiles between 0 and %.1f us, ended between %.1f and %.1f us; lifetimes (tiles only): median %.1f, max %.1f us" % (
    len(marks), t_begin.max(), t_end.min(), t_end.max(), float(np.median(t_end - t_begin)), float((t_end - t_begin).max())))
for at in range(0, int(t_end.max()) + 1, 20):
    alive = int(((t_begin <= at) & (t_end > at)).sum())
    print("  t = %3d us: %4d workgroups in their tiles (%.1f per CU), %4d not begun" % (at, alive, alive / 256.0, int((t_begin > at).sum())))
print("distinct (XCC, SE, CU) triples: %d" % len(set(marks[:, 2].tolist())))
names = {1: "steady", 2: "two binades, regular pixel", 3: "two binades, irregular pixel", 4: "per-pixel loop", 5: "not tiled", 6: "window leaves the frame"}
for k in np.unique(out[..., 0]):
    print("%-32s %9d pixels  %.3f" % (names.get(int(k), k), int((out[..., 0] == k).sum()), float((out[..., 0] == k).mean())))
t = out[..., 1].astype(float) * 0.64
for k in np.unique(out[..., 0]):
    m = out[..., 0] == k
    print("%-32s time per tile (us): median %.1f  max %.1f" % (names.get(int(k), k), float(np.median(t[m])), float(t[m].max())))
tiles = t.reshape(H // 8, 8, W // 32, 32).max(axis=(1, 3))
print("sum of tile times %.0f us over %d tiles; slowest tiles:" % (tiles.sum(), tiles.size))
order = np.argsort(tiles.reshape(-1))[::-1][:8]
for o in order:
    print("  tile row %d column %d: %.1f us (run %d)" % (o // (W // 32), o % (W // 32), tiles.reshape(-1)[o], out[(o // (W // 32)) * 8, (o % (W // 32)) * 32, 2]))
if STATIC:
    groups = tiles.size // 8          # (AO_TILES_PER_GROUP = 8, grid-stride: workgroup g renders tiles g, g + groups, ...)
    per_group = tiles.reshape(-1)[:groups * 8].reshape(8, groups).sum(axis=0)
    print("workgroups: %d; sum of their tiles' times: median %.1f us, 99th percentile %.1f, max %.1f (workgroup %d)" % (
        groups, float(np.median(per_group)), float(np.percentile(per_group, 99)), float(per_group.max()), int(per_group.argmax())))
    slowest = int(per_group.argmax())
    print("  its tiles (row, column, path of the first pixel, us):", [(int((slowest + r * groups) // (W // 32)), int((slowest + r * groups) % (W // 32)),
          int(out[((slowest + r * groups) // (W // 32)) * 8, ((slowest + r * groups) % (W // 32)) * 32, 0]), round(float(tiles.reshape(-1)[slowest + r * groups]), 1)) for r in range(8)])
per_run = [float(t[out[..., 2] == r].mean()) for r in range(8)]
print("mean per-pixel tile time by run index:", [round(v, 1) for v in per_run])
