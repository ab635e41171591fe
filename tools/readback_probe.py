"""Development aid: what the pipelined image read-back (solr_hip_d2h_image_async) costs per frame for a few
combinations of engine frames in flight and host lag.  usage: python tools/readback_probe.py [--scene cornell]"""
import argparse, ctypes as C, importlib, os, sys, time
from collections import deque
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell")
ap.add_argument("--frames", type=int, default=300)
a = ap.parse_args()
hip = solr.hip_lib()
k = solr.Kernel(engine="hip", deterministic_seed=1)
kw = dict(width=1920, height=1080)
if a.scene == "cornell":
    kw["iterations"] = 3
getattr(solr.scenes, a.scene)(k, **kw)
k.L.SolRx_Render(0.0)
flat = k.flat_scene()
si, ppi, eye, direction, angles = k.frame_parameters()
objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
render = lambda: hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
for _ in range(60):
    render()
hip.solr_hip_synchronize()
for flights in (1, 2, 3, 4):
    hip.solr_hip_set_frames_in_flight(flights)
    for lag in range(0, 4):
        tickets = deque()
        def step():
            render()
            tickets.append(hip.solr_hip_d2h_image_async())
            while len(tickets) > lag:
                hip.solr_hip_image_wait(tickets.popleft())
        for _ in range(20):
            step()
        t0 = time.perf_counter()
        for _ in range(a.frames):
            step()
        while tickets:
            hip.solr_hip_image_wait(tickets.popleft())
        dt = (time.perf_counter() - t0) / a.frames
        print("flights %d, host takes the image %d frames back: %.4f ms per frame" % (flights, lag, dt * 1e3), flush=True)
    # no read-back at all, same flights
    t0 = time.perf_counter()
    for _ in range(a.frames):
        render()
    hip.solr_hip_synchronize()
    print("flights %d, no read-back: %.4f ms per frame" % (flights, (time.perf_counter() - t0) / a.frames * 1e3), flush=True)
k.check(0, "probe")
k.finalize()
