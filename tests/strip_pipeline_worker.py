"""Worker of tests/test_strip_pipeline_gpu.py (its own process: see run()).

The N > 1 frame loop of bench.py (sol-r_amd.StripPipeline) on one GPU: an RCCL process group of one
rank, strips gathered to rank 0, one and two frames in flight.  What a multi-GPU run adds to this is
only the size of the group: the event chaining between the engine's streams and the gather stream, the
buffer alternation and the in-order collective are all exercised here against real RCCL."""
import ctypes as C
import importlib
import os

import sys

import numpy as np



def run(flights):
    # torch BEFORE the engine library: importing torch into a process in which another copy of the
    # HIP runtime is already initialised hangs (bench.py does the same)
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import scenes_extra as X
    from helpers import gpu_frame
    solr = importlib.import_module("sol-r_amd")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    hip = solr.hip_lib()
    W, H = 136, 72
    k = solr.Kernel(engine="hip")
    X.sticks(k, width=W, height=H)
    try:
        # two reference frames (cameras A and B) rendered the plain way
        pp, ids, rgb_a = gpu_frame(k)
        k.check(0, "reference frame A")
        flat = k.flat_scene()
        si, ppi, eye, direction, angles = k.frame_parameters()
        si.pathTracingIteration = 0
        eye_b = eye.copy()
        eye_b[0] += 900.0
        objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye_b), fp(direction), fp(angles))
        rgb_b = np.zeros((H, W, 3), np.uint8)
        hip.solr_hip_d2h(C.byref(si), C.c_void_p(rgb_b.ctypes.data), None)
        assert not np.array_equal(rgb_a, rgb_b)

        pipe = solr.StripPipeline(dist, torch, hip, W, H, 0, 1, local_rank=0, frames_in_flight=flights)
        assert hip.solr_hip_get_frames_in_flight() == flights
        cameras = [eye, eye_b, eye_b, eye, eye, eye_b, eye, eye_b, eye_b]
        for n, cam in enumerate(cameras):
            i = pipe.frame(lambda: hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(cam),
                                                       fp(direction), fp(angles)))
            assert i == n
            if n >= 1 and n % 3 == 0:      # look at a frame while later ones are queued
                img = pipe.image(n).cpu().numpy()
                assert np.array_equal(img, rgb_b if cam is eye_b else rgb_a), n
        k.check(0, "pipeline")
        last = pipe.image(len(cameras) - 1).cpu().numpy()
        assert np.array_equal(last, rgb_b)
        if flights > 1:
            before = pipe.image(len(cameras) - 2).cpu().numpy()
            assert np.array_equal(before, rgb_b)   # frame 7 (camera B) is still in its slot
        pipe.drain()
    finally:
        k.finalize()
        dist.destroy_process_group()
    print("STRIP_PIPELINE_OK", flights)


if __name__ == "__main__":
    run(int(sys.argv[1]))
