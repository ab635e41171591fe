"""Development aid: random mirror terrains (long triangle lists, sheets a few per cent of the bounce distance apart, grazing
cameras, ray epsilons from 0.001 to 0.45, view distances that put hits in the band under the initial bound): the frame with
bounce rays on the order-free lists, checked (solr_hip_set_short_ray_lists(1), rt_device.h closestHitWalk), against the frame
with them in the reference's order - bit for bit - and every fourth scene against the CPU oracle as well.
    python tools/fuzz_short_rays.py [first seed] [scenes]"""
import ctypes as C, importlib, math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
solr = importlib.import_module("sol-r_amd")
from oracle import loader
from helpers import assert_parity_pinned, gpu_frame, oracle_frame
S = solr.scenes
hip = solr.hip_lib()
W, H = 160, 120


def build(k, seed):
    rng = S.LCG(seed)
    u = lambda a, b: rng.uniform(a, b)
    pick = lambda seq: seq[rng.next() % len(seq)]
    n, layers = 20 + rng.next() % 24, 1 + rng.next() % 4
    gap = pick([5.0, 20.0, 60.0, 150.0, 400.0])
    k.initialize(width=W, height=H, nbRayIterations=2 + rng.next() % 4, rayEpsilon=pick([0.05, 0.05, 0.001, 0.2, 0.45]),
                 viewDistance=pick([50000.0, 50000.0, 30000.0, 18000.0]), doubleSidedTriangles=int(rng.next() % 4 == 0))
    mats = [k.add_material(u(0.2, 1), u(0.2, 1), u(0.2, 1), reflection=pick([0.0, 0.5, 0.8, 0.9]),
                           transparency=(0.6 if rng.next() % 9 == 0 else 0.0), refraction=1.1, specValue=u(0, 1),
                           specPower=pick([20.0, 100.0])) for _ in range(5)]
    amp, fx, fz = u(100, 1500), u(3, 14), u(3, 14)
    def pt(i, j, layer):
        a, b = i / n - 0.5, j / n - 0.5
        return (a * 20000.0, amp * math.sin(fx * a + 0.4) * math.cos(fz * b) - 2500.0 - gap * layer, b * 20000.0)
    for layer in range(layers):
        for i in range(n):
            for j in range(n):
                a, b, c, d = pt(i, j, layer), pt(i + 1, j, layer), pt(i + 1, j + 1, layer), pt(i, j + 1, layer)
                m = mats[(i // 3 + j // 2 + layer) % 5]
                t = k.add_primitive(solr.ptTriangle, a, b, c, material=m)
                k.set_normals(t, (0, 1, 0), (0.1, 1, 0), (0.1, 1, 0.1))
                t = k.add_primitive(solr.ptTriangle, a, c, d, material=m)
                k.set_normals(t, (0, 1, 0), (0.1, 1, 0.1), (0, 1, 0.1))
    for _ in range(rng.next() % 4):      # a few mirror balls over it
        k.add_primitive(solr.ptSphere, (u(-6000, 6000), u(-2000, 1000), u(-4000, 6000)), size=(u(300, 1500), 0, 0),
                        material=mats[rng.next() % 5])
    S.add_light(k)
    k.compact_boxes(True)
    k.set_camera((u(-3000, 3000), u(-2200, 3000), u(-15000, -9000)), look_at=(u(-2000, 2000), u(-3500, -1500), u(-2000, 4000)))


def frame(seed, short):
    hip.solr_hip_set_short_ray_lists(short)
    k = solr.Kernel(engine="hip")
    build(k, seed)
    for _ in range(2):
        k.render()
    pp, ids, rgb = gpu_frame(k)
    k.check(0, "render")
    return k, (np.array(pp, copy=True), np.array(ids, copy=True), np.array(rgb, copy=True))


first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
hip.solr_hip_set_tile_scheduling(0)
for seed in range(first, first + count):
    k, a = frame(seed, 1)
    deep = hip.solr_hip_order_free_nodes() > 1024
    note = ""
    if seed % 4 == 0:
        misround = np.zeros((H, W), np.uint8)
        opp, oids, orgb, _, status = oracle_frame(k, loader, misround=misround)
        try:
            res = assert_parity_pinned(a, (opp, oids, orgb), misround, 4, "seed %d" % seed)
            note = " oracle ok (%d counted)" % res["pixels_outside_the_bar"]
        except AssertionError as e:
            bad += 1
            note = " ORACLE DIFFERS: %s" % (str(e)[:300])
    k.finalize()
    k, b = frame(seed, 0)
    k.finalize()
    same = np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    if not same:
        bad += 1
        where = np.argwhere((a[1] != b[1]).any(axis=-1) | (a[0].view(np.uint32) != b[0].view(np.uint32)).any(axis=-1))
        note += " FRAMES DIFFER at %d pixels, first %s" % (len(where), where[:4].tolist())
    print("seed %3d deep %d bounced %.2f hit %.2f%s" % (seed, deep, float((a[1][..., 1] > 1).mean()), float((a[1][..., 0] >= 0).mean()), note), flush=True)
hip.solr_hip_set_short_ray_lists(-1)
print("%d scenes, %d outside" % (count, bad))
sys.exit(1 if bad else 0)
