"""Render a scene on the GPU and save the raw buffers under gpurun_out/ for offline
comparison with the oracle (development aid).  usage: dump_frame.py <scene> <w> <h> <it> [k=v ...]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
solr = importlib.import_module("sol-r_amd")
import scenes_extra as X
name = sys.argv[1]; w = int(sys.argv[2]); h = int(sys.argv[3]); it = int(sys.argv[4])
extra = {a.split("=")[0]: eval(a.split("=")[1]) for a in sys.argv[5:]}
k = solr.Kernel(engine="hip")
builder = getattr(solr.scenes, name, None) or getattr(X, name)
builder(k, width=w, height=h, iterations=it, **extra)
rgb = k.render()
pp = k.postprocessing_buffer()
ids = k.primitive_ids()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "frame_%s_%dx%d_%d.npz" % (name, w, h, it)), rgb=rgb, pp=pp, ids=ids)
print("saved", name, rgb.shape)
