#!/bin/bash
# cfg4 under rocprofv3 --stats with each library: mean duration of k_ambientOcclusion and ms per delivered pass
export TMPDIR=/tmp; R=$PWD; LIB=sol-r_amd/csrc/libsolr_hip.so; cp $LIB /tmp/keep.so
for round in 1 2; do
for l in "$@"; do
  cp $l $LIB
  SOLR_BENCH_REGIONS=5 python bench.py --config cfg4 --steps 74 --warmup 12 --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $l)', 'ms per delivered pass', d['ms_per_step'], 'renderer', d['roofline']['kernel_ms'])"
  rm -rf $R/gpurun_out/ao_ab; cd /tmp
  SOLR_BENCH_REGIONS=3 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ao_ab -o t -- python3 $R/bench.py --config cfg4 --steps 74 --warmup 2 --no-cpu-baseline --no-check > /dev/null 2>&1
  cd $R
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/ao_ab/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    if "ambient" in row["Name"]:
        print("   $(basename $l) k_ambientOcclusion", row["Calls"], "calls, mean %.1f us, min %.1f, max %.1f" % (float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
done
done
cp /tmp/keep.so $LIB
