"""Development aid: k_ambientOcclusion (and k_depthOfField, k_default) on random synthetic frame buffers against the
oracle's post-processing, every pixel of the image - frame sizes from one tile to a few hundred tiles (the tile order
with heavy tile rows / columns first, tiles in two binades, irregular pixels taken together, windows that leave the
frame or do not fit LDS), taps of a fraction of a pixel up to hundreds of pixels.
    python tools/fuzz_ao.py [first seed] [count]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
solr = importlib.import_module("sol-r_amd")
import engine_probes as E
from oracle import probes, loader

first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 60)
L = loader.lib()
assert L.oracle_get_dialect() == 0
bad = 0
kinds = {}
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    W = int(rng.choice([rng.integers(1, 40), rng.integers(30, 300), rng.integers(250, 1100), 32 * rng.integers(1, 20), 2048 + rng.integers(-40, 40)]))
    H = int(rng.choice([rng.integers(1, 20), rng.integers(8, 200), rng.integers(150, 700), 8 * rng.integers(1, 40), 1024 + rng.integers(-20, 20)]))
    if W * H > 1100 * 700:
        H = max(1, 1100 * 700 // W)
    n = W * H
    pp = np.zeros(n, solr.PP_DTYPE)
    pp["colorInfo"][:, :3] = rng.uniform(0.0, 1.4, (n, 3)).astype(np.float32)
    ys, xs = np.mgrid[0:H, 0:W]
    depth = 6000.0 + 3000.0 * ((xs // int(rng.integers(3, 40)) + ys // int(rng.integers(2, 30))) % 3) + rng.uniform(-80, 80, (H, W))
    pp["colorInfo"][:, 3] = depth.reshape(-1).astype(np.float32)
    scale = float(rng.choice([0.005, 0.005, 0.02, 1.0]))
    randoms = (scale * rng.uniform(-1.0, 1.0, max(n, 4096))).astype(np.float32)
    if rng.integers(0, 4) == 0:
        randoms = (0.000005 * (rng.integers(0, 2000, len(randoms)) - 1000)).astype(np.float32)   # the host's own distribution
    effect = int(rng.choice([2, 2, 2, 1, 0]))
    param2 = float(rng.choice([10.0, rng.uniform(0.5, 50.0), rng.uniform(50.0, 3000.0)])) if effect == 2 else float(rng.uniform(10.0, 400.0))
    iteration = int(rng.choice([0, 3, 11, 30]))
    si = probes._scene_info(size_x=W, size_y=H, pathTracingIteration=iteration)
    ppi = solr.PostProcessingInfo(effect, float(rng.uniform(2000.0, 12000.0)), param2, int(rng.integers(1, 20)))
    case = dict(name="post", si=si, ppi=ppi, pp=pp, randoms=randoms, width=W, height=H)
    want = probes._oracle_outputs(L, case)["bitmap"]
    got = E.engine_outputs(solr, case)["bitmap"]
    kinds[effect] = kinds.get(effect, 0) + 1
    if not np.array_equal(want, got):
        bad += 1
        d = np.flatnonzero((want.reshape(-1, 3) != got.reshape(-1, 3)).any(axis=1))
        print("seed %d: %d x %d, effect %d, param2 %.3f, randoms x %.4g, pass %d: %d pixels differ, first at (%d, %d)" % (
            seed, W, H, effect, param2, scale, iteration, len(d), d[0] % W, d[0] // W), flush=True)
print("fuzz_ao: %d buffers (%s), %d with a pixel that differs" % (count, ", ".join("%d of effect %d" % (v, k) for k, v in sorted(kinds.items())), bad))
