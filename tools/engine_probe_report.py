"""Engine probes against the reference's probe outputs: where do they agree, key by key (exploration tool; the
assertions live in tests/test_engine_probes_gpu.py).  Needs a GPU.  Prints one JSON line per case."""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SOLR_HIP_FREE_AFTER", "1")

solr = importlib.import_module("sol-r_amd")
from oracle import loader, probes  # noqa: E402
import engine_probes as E  # noqa: E402

fixture = np.load(os.path.join(ROOT, "tests", "golden", "reference_probes.npz"))


def load(name):
    inputs = {k.split("/", 2)[2]: fixture[k] for k in fixture.files if k.startswith(name + "/in/")}
    case = probes.unpack(inputs)
    reference = {}
    for k in fixture.files:
        parts = k.split("/")
        if parts[0] == name and parts[1] != "in":
            reference.setdefault(parts[1], {})[parts[2]] = fixture[k]
    return case, reference


def rows_differ(a, b):
    a, b = np.asarray(a), np.asarray(b)
    same = probes.same_bits(a, b)
    return ~(same.reshape(len(same), -1).all(axis=1))


def main():
    L = loader.lib()
    for name in probes.CASES:
        case, reference = load(name)
        if case["name"] not in E.ENGINE_CASES:
            continue
        variants = [("default", 0, 0)]
        if case["name"] in ("closest", "shadow"):
            variants += [("exact list", 0, 1), ("everything", E.EVERYTHING, 0)]
        if case["name"] == "primitive":
            variants += [("everything", E.EVERYTHING, 0)]
        for label, features, exact in variants:
            try:
                out = E.engine_outputs(solr, case, features=features, exact=exact)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"case": name, "variant": label, "error": str(e)}))
                continue
            ref = reference["source_order"]
            cuda = probes._oracle_outputs(L, case)           # the oracle in dialect 0
            line = {"case": name, "variant": label, "features": int(out.pop("features", 0)), "n": int(len(next(iter(out.values()))))}
            hit_key = "hit" if "hit" in ref else None
            both_hit = None
            if hit_key:
                both_hit = (np.asarray(out["hit"]) != 0) & (np.asarray(ref["hit"]) != 0)
            types = case["prims"]["type"] if "prims" in case and case["name"] in ("primitive", "intersection_shader") else None
            for key, value in out.items():
                rk = "hit" if key.startswith("hit") else key
                if rk not in ref:
                    continue
                d_ref = rows_differ(value, ref[rk])
                d_cuda = rows_differ(value, cuda[rk]) if rk in cuda else None
                if both_hit is not None and not key.startswith("hit") and key != "primitive":
                    d_ref = d_ref & both_hit
                    if d_cuda is not None:
                        d_cuda = d_cuda & (np.asarray(cuda["hit"]) != 0) & (np.asarray(out["hit"]) != 0)
                entry = {"vs_reference": int(d_ref.sum()), "vs_oracle_cuda": int(d_cuda.sum()) if d_cuda is not None else None}
                if types is not None and d_ref.any():
                    entry["by_type_vs_reference"] = {int(t): int(d_ref[types == t].sum()) for t in np.unique(types[d_ref])}
                if types is not None and d_cuda is not None and d_cuda.any():
                    entry["by_type_vs_oracle"] = {int(t): int(d_cuda[types == t].sum()) for t in np.unique(types[d_cuda])}
                if case["name"] == "box" and d_ref.any():
                    zero = (case["directions"] == 0).any(axis=1)
                    entry["with_zero_component"] = int((d_ref & zero).sum())
                    entry["t0_nonzero"] = int((d_ref & (case["t0"] != 0)).sum())
                line[key] = entry
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
