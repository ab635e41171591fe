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
out = E.engine_outputs(solr, case)["bitmap"].reshape(H, W, 3)
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
per_run = [float(t[out[..., 2] == r].mean()) for r in range(8)]
print("mean per-pixel tile time by run index:", [round(v, 1) for v in per_run])
