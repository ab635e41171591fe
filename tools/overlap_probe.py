"""Development aid: how much frame time would two frames in flight save?  Alternates the engine between two
HIP streams (same output buffers: the frames are identical here, so the overlap is harmless)."""
import os, sys, time, importlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
scene = sys.argv[1] if len(sys.argv) > 1 else "cornell"
k = solr.Kernel(engine="hip")
kw = dict(width=1920, height=1080)
if scene == "cornell":
    kw["iterations"] = 3
getattr(solr.scenes, scene)(k, **kw)
k.render()
flat = k.flat_scene(); si, ppi, eye, direction, angles = k.frame_parameters(); si.pathTracingIteration = 0
objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(n, nstreams):
    for i in range(n):
        if nstreams:
            hip.solr_hip_set_stream(C.c_void_p(streams[i % nstreams].cuda_stream))
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
for ns in (1, 2, 1, 2):
    run(60, ns); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(100, ns); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("%s: %d stream(s): %.4f ms/frame" % (scene, ns, (t1 - t0) / 100 * 1e3))
