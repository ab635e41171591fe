"""Development aid: what cost-balanced strips are worth for an eight-rank frame, measured on ONE GPU.  Every
rank's strip is rendered here in turn (no gather): with the equal strips of solr_hip_strip_rows, then - from the
rows' costs those frames recorded, summed as the ranks' all-reduce would - with the strips of
solr_hip_balanced_strips.  An N-GPU frame is as slow as its slowest rank: compare the maxima.
    python tools/strip_balance.py [scene] [world]"""
import os, sys, time, importlib, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = 1920, 1080
FLIGHTS = int(os.environ.get("STRIP_FLIGHTS", "3"))


def strip_time(first, count, costs=None):
    """ms per frame of the strip with three frames in flight; adds its rows' costs to `costs`"""
    if count <= 0:
        return 0.0
    k = solr.Kernel(engine="hip", device=0)
    kw = dict(width=W, height=H)
    if scene == "cornell":
        kw["iterations"] = 3
    getattr(solr.scenes, scene)(k, **kw)
    hip.solr_hip_set_strip(first, count)
    k.L.SolRx_Render(0.0); k.check(0, "first")
    flat = k.flat_scene(); si, ppi, eye, direction, angles = k.frame_parameters()
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    hip.solr_hip_set_frames_in_flight(FLIGHTS)
    for _ in range(40):
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
    hip.solr_hip_synchronize()
    t0 = time.perf_counter()
    n = 400
    for _ in range(n):
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
    hip.solr_hip_synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    if costs is not None:
        costs += solr.strip_row_costs(H)
    hip.solr_hip_set_frames_in_flight(1)
    hip.solr_hip_set_strip(0, -1)
    k.finalize()
    return ms


costs = np.zeros(H, np.float32)
equal = [solr.strip_rows(r, world, H)[:2] for r in range(world)]
t_equal = [strip_time(f, c, costs) for f, c in equal]
balanced = solr.balanced_strips(costs, world)
t_balanced = [strip_time(f, c) for f, c in balanced]
full = strip_time(0, H)
print("%s 1920x1080, %d ranks, %d frames in flight, one rank at a time on one GPU (no gather); whole frame on one GPU %.4f ms" % (
    scene, world, FLIGHTS, full))
print("  equal strips    %s" % "  ".join("%d+%d" % s for s in equal))
print("    ms per strip  %s   slowest %.4f ms  -> %.2f x the single GPU" % (
    "  ".join("%.4f" % t for t in t_equal), max(t_equal), full / max(t_equal)))
print("  balanced strips %s" % "  ".join("%d+%d" % s for s in balanced))
print("    ms per strip  %s   slowest %.4f ms  -> %.2f x the single GPU" % (
    "  ".join("%.4f" % t for t in t_balanced), max(t_balanced), full / max(t_balanced)))
