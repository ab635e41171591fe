"""profiles/<round>/dialect_sites.json: for every dialect switch of oracle/solr_oracle.c (DIALECT(n)) its line, how often
the cases of tests/cuda_text_cases.py evaluated it in dialect 0, and which case notices when that one switch is made
to read the OpenCL form.  CPU only (the cases run the oracle's counting build and the CUDA text model).

    python tools/dialect_sites.py r3        (then: python tools/design_tables.py r3)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cuda_text_cases.py")], capture_output=True, text=True, cwd=ROOT)
if res.returncode != 0:
    sys.exit(res.stderr[-3000:])
d = json.loads(res.stdout.strip().splitlines()[-1])
lines = {}
for no, text in enumerate(open(os.path.join(ROOT, "oracle", "solr_oracle.c")), 1):
    for m in re.finditer(r"DIALECT\((\d+)\)", text):
        lines.setdefault(int(m.group(1)), no)
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
out = {"commit": commit, "cases": len(d["cases"]), "all_same": all(c["same"] for c in d["cases"].values()),
       "sites": [{"site": f["site"], "oracle_line": lines.get(f["site"]), "hits_in_dialect_0": d["site_hits"][f["site"]],
                  "flip_noticed_by": f["noticed_by"]} for f in d["flipped"]]}
path = os.path.join(ROOT, "profiles", tag, "dialect_sites.json")
json.dump(out, open(path, "w"), indent=1)
print("%s: %d cases (all the same bits: %s), %d switches, %d flips unnoticed" % (
    path, out["cases"], out["all_same"], len(out["sites"]), sum(1 for s in out["sites"] if not s["flip_noticed_by"])))
