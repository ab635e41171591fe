#!/bin/bash
# Development aid: per-frame instruction counters of the renderer for a few
# (scene, graphics level, bounces) points.  Run on the GPU box from the repo root:
#   bash tools/pmc_ablate.sh TAG "scene:level:iterations ..."
TAG=$1; shift
POINTS=${1:-"cornell:4:3"}
export TMPDIR=/tmp
ROOT=$PWD
for p in $POINTS; do
  IFS=: read scene level it <<< "$p"
  out=$ROOT/gpurun_out/pmc_${TAG}_${scene}_${level}_${it}
  rm -rf $out
  (cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_BUSY_CYCLES \
      -d $out -o pmc -- python3 $ROOT/bench.py --scene $scene --graphics-level $level --iterations $it --steps 3 --warmup 1 --no-cpu-baseline > $out.log 2>&1)
  echo "== $p"; tail -1 $out.log | cut -c1-200
  python3 $ROOT/tools/pmc_summary.py "$out/**/*counter_collection.csv" "k_standardRenderer<0,"
done
