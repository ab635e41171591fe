"""Development aid: host-side cost of each call of the N > 1 frame loop, measured with one rank on RCCL."""
import os, sys, time, importlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_stream(torch.cuda.Stream())
W, H = 1920, int(sys.argv[1]) if len(sys.argv) > 1 else 135
k = solr.Kernel(engine="hip", device=0)
solr.scenes.cornell(k, width=W, height=H, iterations=3)
hip.solr_hip_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))
sg = solr.StripGather(dist, torch, W, H, 0, 1, device="cuda")
hip.solr_hip_bind_device_bitmap(C.c_void_p(sg.buffer(0).data_ptr()))
k.L.SolRx_Render(0.0); k.check(0, "first")
flat = k.flat_scene(); si, ppi, eye, direction, angles = k.frame_parameters()
objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
T = {"buffer": 0.0, "render": 0.0, "submit": 0.0, "event": 0.0}
def run(n, gather=True, throttle=True):
    ev = [None, None]
    for k_ in T: T[k_] = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        if throttle and ev[i % 2] is not None: ev[i % 2].synchronize()
        b = time.perf_counter()
        buf = sg.buffer(i) if gather else sg.strips[i % 2]
        hip.solr_hip_bind_device_bitmap(C.c_void_p(buf.data_ptr()))
        c = time.perf_counter()
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        d = time.perf_counter()
        if throttle:
            ev[i % 2] = torch.cuda.Event(); ev[i % 2].record()
        e = time.perf_counter()
        if gather: sg.submit(i)
        f = time.perf_counter()
        T["event"] += (b - a) + (e - d); T["buffer"] += c - b; T["render"] += d - c; T["submit"] += f - e
    issued = time.perf_counter()
    sg.drain(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("gather=%d throttle=%d: %.1f us/frame total, issue %.1f us/frame; host per call (us): %s" % (
        gather, throttle, (t1 - t0) / n * 1e6, (issued - t0) / n * 1e6, {k_: round(v / n * 1e6, 1) for k_, v in T.items()}))
def run_sync(n):
    """gather issued with async_op=False on one buffer"""
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        hip.solr_hip_bind_device_bitmap(C.c_void_p(sg.strips[0].data_ptr()))
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        dist.gather(sg.strips[0], sg.lists[0], dst=0)
    issued = time.perf_counter()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("gather sync: %.1f us/frame total, issue %.1f us/frame" % ((t1 - t0) / n * 1e6, (issued - t0) / n * 1e6))
for g_, t_ in ((True, True), (False, False)):
    run(20, g_, t_); run(200, g_, t_)
run_sync(20); run_sync(200); run_sync(200)
dist.destroy_process_group()
