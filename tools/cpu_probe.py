"""How does the CPU oracle scale with threads on this box? (development aid)"""
import importlib, sys, time, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
solr = importlib.import_module("sol-r_amd")
from oracle import loader
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread' | head -5")
k = solr.Kernel(engine="host-only")
solr.scenes.cornell(k, width=1920, height=1080, iterations=3)
fs = k.flat_scene(); si, pp, eye, d, ang = k.frame_parameters()
L = loader.lib(); s = loader.Scene(fs)
rows = 1080; w = 1920
ppb = np.zeros((rows, w, 8), np.float32); ids = np.zeros((rows, w, 4), np.int32); bmp = np.zeros((rows, w, 3), np.uint8)
counts = (C.c_ulonglong * 4)()
for nt in (1, 8, 16, 32, 64, 128, 256, 64, 32):
    best = 1e9
    for rep in range(2):
        ppb[:] = 0; ids[:] = 0
        t = time.time()
        L.oracle_render(C.byref(s.c), C.addressof(si), C.addressof(pp), eye.ctypes.data, d.ctypes.data, ang.ctypes.data, 0, rows, ppb.ctypes.data, ids.ctypes.data, bmp.ctypes.data, C.addressof(counts), nt)
        best = min(best, time.time() - t)
    print(nt, "threads", round(best, 4), "s", round((counts[0] + counts[1]) / best / 1e6, 2), "Mrays/s", flush=True)
