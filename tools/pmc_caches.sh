#!/bin/bash
# Development aid: instruction-cache and scalar-data-cache behaviour of the renderer kernel (rocprofv3 --pmc, two passes).
#   bash tools/pmc_caches.sh [scene]      (on the GPU box, from the repo root)
SCENE=${1:-cornell}
ROOT=$PWD; OUT=$ROOT/gpurun_out/caches_$SCENE; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --scene $SCENE --steps 24 --warmup 12 --no-cpu-baseline --frames-in-flight 1"
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    --output-format csv -d $OUT/i -o pmc -- $CMD > $OUT/i.log 2>&1
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/d -o pmc -- $CMD > $OUT/d.log 2>&1
cd $ROOT
python3 - <<PY
import csv, glob, collections
for p in ("i", "d"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % p):
        for r in csv.DictReader(open(f)):
            if "k_standardRenderer<0," in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print("%-30s %16.0f   (mean of %d launches)" % (k, sum(agg[k]) / len(agg[k]), len(agg[k])))
PY
