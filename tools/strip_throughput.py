"""Development aid: what ONE rank of an N-GPU run sustains in the renderer - frames of its row strip per
millisecond with 1 ... 4 frames in flight, on one GPU (no gather).  A 1/8 strip of the 1080p frame is one
round of waves: its frame takes as long as its longest wave unless several frames overlap.
    python tools/strip_throughput.py [scene] [all]      # all: every rank of the eight, not ranks 0, 4 and 7"""
import os, sys, time, importlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W, H = 1920, 1080
for world in (1, 8):
    for rank in (range(world) if "all" in sys.argv[2:] else sorted(set([0, world // 2, world - 1]))):
        k = solr.Kernel(engine="hip", device=0)
        kw = dict(width=W, height=H)
        if scene == "cornell":
            kw["iterations"] = 3
        getattr(solr.scenes, scene)(k, **kw)
        first, count, per = solr.strip_rows(rank, world, H)
        hip.solr_hip_set_strip(first, count)
        k.L.SolRx_Render(0.0); k.check(0, "first")
        flat = k.flat_scene(); si, ppi, eye, direction, angles = k.frame_parameters()
        objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        out = []
        for flights in (1, 2, 3, 4):
            hip.solr_hip_set_frames_in_flight(flights)
            for _ in range(40):
                hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
            hip.solr_hip_synchronize()
            t0 = time.perf_counter()
            n = 400
            for _ in range(n):
                hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
            hip.solr_hip_synchronize()
            out.append((time.perf_counter() - t0) / n * 1e3)
        hip.solr_hip_set_frames_in_flight(1)
        k.finalize()
        print("%s world %d rank %d (rows %d..%d): ms per strip frame with 1-4 frames in flight: %s" % (
            scene, world, rank, first, first + count - 1, "  ".join("%.4f" % t for t in out)))
