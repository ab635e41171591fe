"""Makes tests/golden/reference_probes.npz: outputs of the REFERENCE'S OWN functions (its OpenCL engine,
compiled from /root/reference by oracle/Makefile into oracle/_ref/, wrapped by oracle/ref_probes.cl) over
the seeded input arrays of oracle/probes.py.  Needs an MI355X (the probes are OpenCL kernels for gfx950):

    gpurun -- 'python tests/golden/make_probe_fixtures.py gpurun_out/reference_probes.npz'
    cp gpurun_out/reference_probes.npz tests/golden/

The file holds, per case, the input arrays (oracle.probes.pack) and what the reference's functions returned
for them, from both builds of the probes; tests/test_reference_probes.py feeds the same inputs to the oracle
on CPU and compares.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import probes  # noqa: E402


def main(out_path):
    if not probes.have_probes():
        raise SystemExit("oracle/_ref is not built (python -c 'import __graft_entry__ as g; g.build()' where "
                         "/root/reference exists)")
    store = {}
    for name, make in probes.CASES.items():
        case = make()
        for key, value in probes.pack(case).items():
            store["%s/in/%s" % (name, key)] = value
        variants = ["renderer"] if case["name"] == "post" else ["source_order", "as_built"]
        for variant in variants:
            out = probes.reference_outputs(case, variant)
            for key, value in out.items():
                store["%s/%s/%s" % (name, variant, key)] = value
        print("%-30s %s" % (name, ", ".join(sorted(out))), flush=True)
    np.savez_compressed(out_path, **store)
    print("wrote %s (%d arrays, %.1f KB)" % (out_path, len(store), os.path.getsize(out_path) / 1024.0))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "reference_probes.npz"))
