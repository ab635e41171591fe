"""Development aid: host-side cost of a scene change (compactBoxes + h2d_scene incl. node-list preparation)."""
import os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
for scene in ("cornell", "height_field", "molecule"):
    k = solr.Kernel(engine="hip")
    getattr(solr.scenes, scene)(k, width=256, height=128)
    t0 = time.perf_counter(); k.compact_boxes(True); t1 = time.perf_counter()
    k.render(); hip.solr_hip_synchronize(); t2 = time.perf_counter()
    k.render(); hip.solr_hip_synchronize(); t3 = time.perf_counter()
    print("%s: compactBoxes(true) %.3f s, first render (h2d_scene + upload + frame) %.3f s, next render %.4f s" % (scene, t1 - t0, t2 - t1, t3 - t2))
    k.finalize()
