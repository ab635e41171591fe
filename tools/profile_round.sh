#!/bin/bash
# Collects the measurements a round commits under profiles/<tag>/ (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r1 [scene]
# 1. bench.py with its defaults (incl. cpu_baseline)        -> bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command    -> kernel_stats.csv
# 3. rocprofv3 --pmc, separate passes (SQ counters; FETCH_SIZE; WRITE_SIZE) -> pmc_*.csv
TAG=${1:-r1}; SCENE=${2:-cornell}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
EXTRA=""; [ "$SCENE" != cornell ] && EXTRA="--no-cpu-baseline"
WHAT="--scene $SCENE"; STEPS=24
# cfg4: BASELINE configs[4], 3840 x 2160 through passes 0...73 (a step is a pass; whole cycles)
[ "$SCENE" = cfg4 ] && { WHAT="--config cfg4"; STEPS=74; }
timeout 900 python bench.py $WHAT $EXTRA > $OUT/bench_$SCENE.json 2> $OUT/bench_$SCENE.err; tail -1 $OUT/bench_$SCENE.json | cut -c1-400
# one frame at a time under the profiler: per-launch durations and counters are then those of the kernel alone
# (bench.py takes roofline.kernel_ms the same way).  Three timed regions are plenty for counters; no walk replay.
export SOLR_BENCH_REGIONS=3
CMD="python3 $ROOT/bench.py $WHAT --steps $STEPS --warmup 12 --no-cpu-baseline --no-walk-bound --frames-in-flight 1"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$SCENE -o trace -- $CMD > $OUT/trace_$SCENE.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_BUSY_CYCLES \
    --output-format csv -d $OUT/pmc_sq_$SCENE -o pmc -- $CMD > $OUT/pmc_sq_$SCENE.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR \
    --output-format csv -d $OUT/pmc_sq2_$SCENE -o pmc -- $CMD > $OUT/pmc_sq2_$SCENE.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$SCENE -o pmc -- $CMD > $OUT/pmc_fetch_$SCENE.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$SCENE -o pmc -- $CMD > $OUT/pmc_write_$SCENE.log 2>&1
cd $ROOT
find $OUT -name "*.csv" | head -20
