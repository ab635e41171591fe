"""Frame time of an animated scene: rotatePrimitives + compactBoxes(false) + render per frame
(apps/scenes/science/MoleculeScene.cpp:75-81), by the host route (rotate and flatten on the host, upload)
and by the device route (solr_hip_rotate_primitives on the resident scene).
usage: python tools/animated_frame.py [--atoms 50000] [--frames 200] [--host-frames 3]"""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="molecule")
ap.add_argument("--atoms", type=int, default=50000)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--host-frames", type=int, default=3)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()

hip = solr.hip_lib()
k = solr.Kernel(engine="hip", deterministic_seed=1)
kw = dict(width=a.width, height=a.height)
if a.scene == "molecule":
    kw["atoms"] = a.atoms
getattr(solr.scenes, a.scene)(k, **kw)
k.render()
step = ((0.0, 0.0, 0.0), (0.0, 0.02, 0.0))

# host route: a touch of the scene store before every rotation keeps it off the device
t = []
for _ in range(a.host_frames):
    k.L.SolR_SetPrimitiveMaterial(0, k.L.SolR_GetPrimitiveMaterial(0))
    t0 = time.perf_counter()
    k.rotate_primitives(*step)
    t1 = time.perf_counter()
    assert k.pending_rotations() == 0
    k.render()
    t2 = time.perf_counter()
    t.append((t1 - t0, t2 - t1))
print("host route  : rotate+flatten %.1f ms, upload+render+readback %.1f ms per frame (%d frames)"
      % (1e3 * sum(x[0] for x in t) / len(t), 1e3 * sum(x[1] for x in t) / len(t), len(t)))

k.rotate_primitives(*step)      # from the last upload on the scene is resident and untouched
k.render()
hip.solr_hip_synchronize()
before = k.pending_rotations()
t0 = time.perf_counter()
rot = 0.0
for _ in range(a.frames):
    r0 = time.perf_counter()
    k.rotate_primitives(*step)
    rot += time.perf_counter() - r0
    k.render()
hip.solr_hip_synchronize()
dt = time.perf_counter() - t0
assert k.pending_rotations() == before + a.frames, k.pending_rotations()
print("device route: %.3f ms per frame, of which rotate+refit %.3f ms (%d frames, %dx%d, %s)"
      % (1e3 * dt / a.frames, 1e3 * rot / a.frames, a.frames, a.width, a.height, a.scene))
t0 = time.perf_counter()
k.sync_host()
print("host store catches up on %d rotations in %.1f ms" % (before + a.frames, 1e3 * (time.perf_counter() - t0)))
k.finalize()
