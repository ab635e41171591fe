"""Compare a GPU dump (tools/dump_frame.py) with the oracle rendered here (development aid)."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
solr = importlib.import_module("sol-r_amd")
from oracle import loader
from helpers import ulp_distance
import scenes_extra as X
name, w, h, it = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
extra = {a.split("=")[0]: eval(a.split("=")[1]) for a in sys.argv[5:]}
k = solr.Kernel(engine="host-only")
builder = getattr(solr.scenes, name, None) or getattr(X, name)
builder(k, width=w, height=h, iterations=it, **extra)
fs = k.flat_scene(); si, pp, eye, d, ang = k.frame_parameters()
opp, oids, obmp, counts, status = loader.render(fs, si, pp, eye, d, ang)
g = np.load(os.path.join(ROOT, 'gpurun_out/frame_%s_%dx%d_%d.npz' % (name, w, h, it)))
gpp, gids, grgb = g['pp'], g['ids'], g['rgb']
u = ulp_distance(gpp[..., :3], opp[..., :3]).max(axis=-1)
absd = np.abs(gpp[..., :3] - opp[..., :3]).max(axis=-1)
print("ids equal", np.array_equal(gids, oids), "rgb equal", np.array_equal(grgb, obmp), "rgb maxdiff", np.abs(grgb.astype(int)-obmp.astype(int)).max())
print("max ulp", u.max(), "max abs diff", absd.max(), "pixels with diff", (u > 0).sum(), ">1ulp", (u > 1).sum())
prims = {int(p['index']): p for p in fs.primitives}
first = oids[..., 0]
for pid in np.unique(first):
    m = first == pid
    t = int(prims[int(pid)]['type']) if pid >= 0 else -1
    mat = int(prims[int(pid)]['materialId']) if pid >= 0 else -1
    print("prim", pid, "type", t, "mat", mat, "n", m.sum(), "diffpix", (u[m] > 1).sum(), "maxabs", absd[m].max())
ys, xs = np.where(u > 1)
for y, x in list(zip(ys, xs))[:6]:
    print(y, x, gpp[y, x, :4], opp[y, x, :4], gids[y, x], oids[y, x])
