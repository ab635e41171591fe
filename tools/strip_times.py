"""Development aid: kernel time of every row strip of the 1080p frame for world = 1, 2, 4, 8 on ONE GPU
(what each rank of an N-GPU run would spend in the renderer)."""
import os, sys, importlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W, H = 1920, 1080
for world in (1, 2, 4, 8):
    times = []
    for rank in range(world):
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
        for _ in range(3):
            hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        hip.solr_hip_synchronize(); hip.solr_hip_kernel_time(None, 1); hip.solr_hip_enable_timing(1)
        for _ in range(20):
            hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        n = C.c_int(0); ms = hip.solr_hip_kernel_time(C.byref(n), 1); hip.solr_hip_enable_timing(0)
        times.append(ms / n.value * 1e3)
        k.finalize()
    print("%s world %d: strip kernel us = %s; max %.1f; ideal %.1f" % (scene, world, [round(t, 1) for t in times], max(times), times and sum(times) / world))
