#!/bin/bash
# Line coverage (gcov) of oracle/solr_oracle.c in dialect 0 under the cases that pin it: tests/cuda_text_cases.py
# (the CUDA text model's frames and function cases) and tests/test_oracle_known_answers.py.
#   bash tools/oracle_coverage.sh > profiles/rN/oracle_coverage.txt        (CPU only; needs gcc + gcov)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=$(mktemp -d)
cp $ROOT/oracle/solr_oracle.c $ROOT/oracle/solr_oracle.h $WORK/
mkdir -p $WORK/include && cp $ROOT/include/solr_types.h $WORK/include/
sed -i 's#"../include/solr_types.h"#"include/solr_types.h"#' $WORK/solr_oracle.h
(cd $WORK && gcc -O0 -g --coverage -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -DORACLE_SITE_COVERAGE -shared -o libsolr_oracle_cov.so solr_oracle.c -lm)
# the cases load oracle/libsolr_oracle_cov.so: put the instrumented build there for the run, restore afterwards
make -s -C $ROOT/oracle coverage
cp $ROOT/oracle/libsolr_oracle_cov.so $WORK/plain_cov.so
cp $WORK/libsolr_oracle_cov.so $ROOT/oracle/libsolr_oracle_cov.so
touch $ROOT/oracle/libsolr_oracle_cov.so
trap "cp $WORK/plain_cov.so $ROOT/oracle/libsolr_oracle_cov.so" EXIT
(cd $ROOT && python tests/cuda_text_cases.py > $WORK/cases.json)
(cd $ROOT && python - <<PY
import sys
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
from oracle import loader
loader.use_coverage_build()
import pytest
sys.exit(pytest.main(["-q", "-x", "$ROOT/tests/test_oracle_known_answers.py", "-p", "no:cacheprovider"]))
PY
) > $WORK/known.log 2>&1 || { tail -20 $WORK/known.log; exit 1; }
cd $WORK && gcov -b -o libsolr_oracle_cov.so-solr_oracle.gcno solr_oracle.c > gcov.log 2>&1
echo "# gcov of oracle/solr_oracle.c (gcc -O0 --coverage), dialect 0 only, under tests/cuda_text_cases.py (every case + one flip"
echo "# of every dialect switch) and tests/test_oracle_known_answers.py; commit $(cd $ROOT && git rev-parse --short HEAD)"
grep -A3 "File 'solr_oracle.c'" gcov.log
python3 - <<'PY'
import re
rows = open("solr_oracle.c.gcov").read().split("\n")
unexecuted = []
for r in rows:
    m = re.match(r"\s*(#####|=====):\s*(\d+):(.*)", r)
    if m:
        unexecuted.append((int(m.group(2)), m.group(3)))
print("lines never executed: %d" % len(unexecuted))
# group into runs for reading
runs, start, prev = [], None, None
for n, _ in unexecuted:
    if start is None:
        start = prev = n
    elif n <= prev + 3:
        prev = n
    else:
        runs.append((start, prev)); start = prev = n
if start is not None:
    runs.append((start, prev))
print("in %d runs (source lines of this commit):" % len(runs))
src = {n: t for n, t in unexecuted}
for a, b in runs:
    first = src[a].strip()[:90]
    print("  %5d-%-5d %s" % (a, b, first))
PY
