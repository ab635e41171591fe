"""Development aid: what a frame's LONGEST tiles spend their time on (the timing build's per-workgroup counters, raster
order, one frame at a time): closest-hit walks with the primary ray's share, the second walks of the checked short-ray
form (rt_device.h closestHitWalk), shadow walks, node loop against leaves.
    make -C sol-r_amd -B EXTRA_HIPFLAGS=-DSOLR_TIMING csrc/libsolr_hip.so && python tools/tile_time_split.py [scene] [short-ray lists 0|1]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
scene = sys.argv[1] if len(sys.argv) > 1 else "height_field"
W, H = 1920, 1080
k = solr.Kernel(engine="hip")
getattr(solr.scenes, scene)(k, width=W, height=H)
hip.solr_hip_set_tile_scheduling(0)
hip.solr_hip_set_frames_in_flight(1)
if len(sys.argv) > 2:
    hip.solr_hip_set_short_ray_lists(int(sys.argv[2]))
if not hasattr(hip, "solr_hip_wave_cycle_slots"):
    sys.exit("not the timing build: make -C sol-r_amd -B EXTRA_HIPFLAGS=-DSOLR_TIMING csrc/libsolr_hip.so")
for _ in range(4):
    k.render()
out = (C.c_ulonglong * 16)()
hip.solr_hip_wave_cycles(out, 1)
k.render()
tiles = 240 * 135
slots = np.zeros((tiles, 16), np.uint64)
hip.solr_hip_wave_cycle_slots.restype = C.c_int
got = hip.solr_hip_wave_cycle_slots(C.c_void_p(slots.ctypes.data), tiles)
s = slots.astype(np.float64)
order = np.argsort(-s[:, 0])[:14]
print("cycles (shader clock) per tile: total | closest (primary) | again time, walks, lanes | shadow | node, leaf | advance first/again | checked walks | nLeaf")
for t in order:
    r = slots[t]
    print("tile (%3d,%3d): %8d | closest %8d (primary %7d) | again %8d walks %d lanes %d | shadow %7d | node %8d leaf %8d | adv %d / %d | checked %d | leaves %d" % (
        t % 240, t // 240, r[0], r[1], r[11], r[12], r[13] >> np.uint64(32), r[13] & np.uint64(0xffffffff), r[2], r[3], r[4],
        r[14] >> np.uint64(32), r[14] & np.uint64(0xffffffff), r[15], r[6]))
tot = slots.sum(axis=0).astype(np.float64)
print("whole frame: total %.3g closest %.1f%% (primary %.1f%%) again %.1f%% shadow %.1f%% node %.1f%% leaf %.1f%%" % (
    tot[0], 100 * tot[1] / tot[0], 100 * tot[11] / tot[0], 100 * tot[12] / tot[0], 100 * tot[2] / tot[0], 100 * tot[3] / tot[0], 100 * tot[4] / tot[0]))
print("again walks %d of %d checked walks, lanes in them %d" % (int(slots[:, 13].sum()) >> 32, int(slots[:, 15].sum()), int((slots[:, 13] & np.uint64(0xffffffff)).sum())))
k.finalize()
