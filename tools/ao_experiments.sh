#!/bin/bash
# Development aid (round 5): where does k_ambientOcclusion spend its time?  Builds of the library with one part of the
# kernel taken out (-DSOLR_AO_EXP=1 no image store, 2 no window gather, 3 no read of the pixel's own record), each
# under rocprofv3 --kernel-trace --stats on one cycle of cfg4's passes.   bash tools/ao_experiments.sh ab/libsolr_hip_ao*.so
R=$PWD; LIB=sol-r_amd/csrc/libsolr_hip.so; cp $LIB /tmp/keep.so
export TMPDIR=/tmp SOLR_BENCH_REGIONS=1
for l in /tmp/keep.so "$@"; do
  cp $l $LIB
  rm -rf $R/gpurun_out/ao_trace; cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ao_trace -o t -- python3 $R/bench.py --config cfg4 --steps 74 --warmup 2 --no-cpu-baseline --no-walk-bound --frames-in-flight 1 --no-check > /dev/null 2>&1
  cd $R
  f=$(find gpurun_out/ao_trace -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$(basename $l)" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if "k_ambientOcclusion" in row["Name"] or "k_standardRenderer<0" in row["Name"]:
        print("%-24s %-40s calls %5s mean %9.1f us" % (sys.argv[2], row["Name"][:40], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
done
cp /tmp/keep.so $LIB
