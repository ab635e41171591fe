"""Compare a GPU dump (tools/dump_frame.py) with the oracle rendered here (development aid)."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
solr = importlib.import_module("sol-r_amd")
from oracle import loader
from helpers import ulp_distance
name, w, h, it = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
k = solr.Kernel(engine="host-only")
getattr(solr.scenes, name)(k, width=w, height=h, iterations=it)
fs = k.flat_scene(); si, pp, eye, d, ang = k.frame_parameters()
opp, oids, obmp, counts, status = loader.render(fs, si, pp, eye, d, ang)
g = np.load(os.path.join(ROOT, 'gpurun_out/frame_%s_%dx%d_%d.npz' % (name, w, h, it)))
gpp, gids, grgb = g['pp'], g['ids'], g['rgb']
u = ulp_distance(gpp[..., :3], opp[..., :3]).max(axis=-1)
absd = np.abs(gpp[..., :3] - opp[..., :3]).max(axis=-1)
print("ids equal", np.array_equal(gids, oids), "rgb equal", np.array_equal(grgb, obmp), "rgb maxdiff", np.abs(grgb.astype(int)-obmp.astype(int)).max())
print("max ulp", u.max(), "max abs diff", absd.max(), "pixels with diff", (u > 0).sum(), ">1ulp", (u > 1).sum())
du = ulp_distance(gpp[..., 3], opp[..., 3])
print("depth ulp", du.max(), (du > 0).sum())
ys, xs = np.where(u > 1)
for y, x in list(zip(ys, xs))[:8]:
    print(y, x, gpp[y, x, :4], opp[y, x, :4], gids[y, x], oids[y, x])
