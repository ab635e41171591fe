"""Development aid: reference OpenCL renderer (oracle/_ref) vs CPU oracle, agreement statistics per scene."""
import os, sys, importlib, argparse
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
solr = importlib.import_module("sol-r_amd")
from oracle import loader
import scenes_extra

def build(spec):
    name, *opts = spec.split(":")
    kw = {}
    for o in opts:
        a, b = o.split("=")
        kw[a] = float(b) if "." in b else int(b)
    k = solr.Kernel(engine="host-only")
    fn = getattr(solr.scenes, name, None) or getattr(scenes_extra, name)
    kw.setdefault("width", 256); kw.setdefault("height", 192)
    fn(k, **kw)
    if name in ("cornell", "height_field", "molecule"):
        k.set_camera((131.0, 77.0, -15000.0), look_at=(57.0, 23.0, 0.0))  # no exactly zero direction component
    return k

for spec in sys.argv[1:]:
    k = build(spec)
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    opp, oids, orgb, counts, status = loader.render(flat, si, ppi, eye, direction, angles, nthreads=8)
    d = np.array(direction, np.float32).copy()
    d[0] -= np.float32(3.0); d[1] -= np.float32(5.0)   # undo RayTracer.cl:2526-2527 (exact when angles == 0)
    rpp, rids, rrgb = loader.ref_render(flat, si, ppi, eye, d, angles)
    same = (oids[..., 0] == rids[..., 0])
    diff = np.abs(orgb.astype(int) - rrgb.astype(int)).max(axis=2)
    dc = np.abs(opp[..., :3] - rpp[..., :3]).max(axis=2)
    rel = dc / np.maximum(np.abs(opp[..., :3]).max(axis=2), 1e-3)
    print("%-44s ids equal %.3f%% | RGB8 ==0 %.2f%% <=1 %.2f%% <=8 %.2f%% max %d | float rel diff: <=1e-5 on %.2f%%, <=1e-3 on %.2f%%, median %.1e" % (
        spec, 100 * same.mean(), 100 * (diff == 0).mean(), 100 * (diff <= 1).mean(), 100 * (diff <= 8).mean(), diff.max(),
        100 * (rel <= 1e-5).mean(), 100 * (rel <= 1e-3).mean(), np.median(rel)))
    np.savez_compressed(os.path.join(ROOT, "gpurun_out/ref_compare_%s.npz" % spec.replace(":", "_").replace("=", "")),
                        orgb=orgb, rrgb=rrgb, oids=oids, rids=rids, opp=opp, rpp=rpp)
