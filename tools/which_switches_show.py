"""Development aid: on which elements of a probe case (oracle/probes.py CASES) does each of the 36 statements in which the
reference's two engines differ SHOW?  The counting build of the oracle flips one switch at a time to the OpenCL arm
(oracle_flip_site) while the others stay CUDA; an output that changes is one that switch can be seen in.  What the
switch-observability mask of tests/test_engine_probes_gpu.py leaves out of "held to the reference's output", by cause.
No GPU needed.
    python tools/which_switches_show.py launch_mesh_100 [more cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import loader  # noqa: E402

loader.use_coverage_build()
from oracle import probes  # noqa: E402

L = loader.lib()
for name in sys.argv[1:]:
    case = probes.CASES[name]()
    L.oracle_flip_site(-1)
    base = probes._oracle_outputs(L, case)
    agree = probes.dialects_agree(case, per_key=True)
    print("%s: dialects agree on %s" % (name, {k: round(float(v.mean()), 4) for k, v in agree.items()}))
    for site in range(36):
        L.oracle_flip_site(site)
        out = probes._oracle_outputs(L, case)
        L.oracle_flip_site(-1)
        shown = {}
        for key in base:
            same = probes.same_bits(base[key], out[key])
            changed = 1.0 - float(same.reshape(len(same), -1).all(axis=1).mean())
            if changed > 0:
                shown[key] = round(changed, 4)
        if shown:
            print("    switch %2d shows on %s" % (site, shown))
