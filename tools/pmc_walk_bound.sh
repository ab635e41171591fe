#!/bin/bash
# Development aid: what binds the node loop?  rocprofv3 --pmc (separate passes) over the walk replay (k_walkBound: a frame's
# walks with nothing but the hand-scheduled node loop, solr_hip_walk_bound) inside bench.py, per launch of that kernel.
#   bash tools/pmc_walk_bound.sh TAG [scene ...]      (GPU box, from the repo root) -> gpurun_out/TAG/pmc_walk_bound_<scene>.txt
TAG=${1:-r6}; shift
SCENES=${@:-height_field molecule cornell}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export SOLR_BENCH_REGIONS=1
cd /tmp
for SCENE in $SCENES; do
  CMD="python3 $ROOT/bench.py --scene $SCENE --steps 8 --warmup 4 --no-cpu-baseline --frames-in-flight 1"
  P=0
  for SET in \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_WAVES SQ_BUSY_CU_CYCLES" \
    "GRBM_GUI_ACTIVE GRBM_COUNT"
  do
    P=$((P+1))
    rm -rf $OUT/wb_${SCENE}_$P
    timeout 600 rocprofv3 --pmc $SET --output-format csv -d $OUT/wb_${SCENE}_$P -o pmc -- $CMD > $OUT/wb_${SCENE}_$P.log 2>&1
  done
  python3 - "$OUT" "$SCENE" <<'PY'
import sys, glob, csv, collections
out, scene = sys.argv[1], sys.argv[2]
for kernel, label in (("k_walkBound", "the walk replay (node loop alone)"), ("k_standardRenderer<0,", "the renderer")):
    agg = collections.defaultdict(list)
    for f in glob.glob("%s/wb_%s_*/**/*counter_collection.csv" % (out, scene), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open("%s/pmc_walk_bound_%s.txt" % (out, scene), "a" if kernel != "k_walkBound" else "w") as o:
        o.write("# %s, %s: per launch\n" % (scene, label))
        for k in sorted(agg):
            o.write("%-26s %16.0f   (mean of %d launches)\n" % (k, sum(agg[k]) / len(agg[k]), len(agg[k])))
print(open("%s/pmc_walk_bound_%s.txt" % (out, scene)).read())
PY
  rm -rf $OUT/wb_${SCENE}_?
done
