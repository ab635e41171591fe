"""Development aid: the molecule scene uploaded three times in one process (new Kernel each time), optionally with glibc malloc
options set first (argument: letters m/M/t/p) - how much of a scene change is page faults of fresh host memory."""
import ctypes, os, sys, time, importlib
libc = ctypes.CDLL("libc.so.6")
M_TRIM_THRESHOLD, M_TOP_PAD, M_MMAP_THRESHOLD = -1, -2, -3
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if "m" in mode:
    print("mmap_threshold 32MB ->", libc.mallopt(M_MMAP_THRESHOLD, 32 << 20))
if "M" in mode:
    print("mmap_threshold 1GB ->", libc.mallopt(M_MMAP_THRESHOLD, 1 << 30))
if "t" in mode:
    print("trim 1GB ->", libc.mallopt(M_TRIM_THRESHOLD, 1 << 30))
if "p" in mode:
    print("top pad 256MB ->", libc.mallopt(M_TOP_PAD, 256 << 20))
sys.path.insert(0, os.getcwd())
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
for scene in ("molecule", "molecule", "molecule"):
    k = solr.Kernel(engine="hip")
    getattr(solr.scenes, scene)(k, width=256, height=128)
    t0 = time.perf_counter(); k.compact_boxes(True); t1 = time.perf_counter()
    k.render(); hip.solr_hip_synchronize(); t2 = time.perf_counter()
    k.render(); hip.solr_hip_synchronize(); t3 = time.perf_counter()
    print("%s %s: compactBoxes(true) %.4f s, first render %.4f s, next render %.4f s  total %.1f ms" % (mode, scene, t1 - t0, t2 - t1, t3 - t2, 1e3 * (t3 - t0)))
    k.finalize()
