"""Render a scene on the GPU and save the raw buffers under gpurun_out/ for offline
comparison with the oracle (development aid)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
solr = importlib.import_module("sol-r_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 160
h = int(sys.argv[3]) if len(sys.argv) > 3 else 120
it = int(sys.argv[4]) if len(sys.argv) > 4 else 1
k = solr.Kernel(engine="hip")
getattr(solr.scenes, name)(k, width=w, height=h, iterations=it)
rgb = k.render()
pp = k.postprocessing_buffer()
ids = k.primitive_ids()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "frame_%s_%dx%d_%d.npz" % (name, w, h, it)), rgb=rgb, pp=pp, ids=ids)
print("saved", rgb.shape, pp.shape)
