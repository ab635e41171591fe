#!/bin/bash
# Development aid: which pipe of the CU is the renderer's busiest?  Separate rocprofv3 --pmc passes
# (8 SQ slots each, --kernel-trace/--stats never combined with --pmc) of bench.py, one frame at a time.
#   bash tools/pmc_pipes.sh TAG [scene ...]        (GPU box, from the repo root)  -> gpurun_out/TAG/pipes_<scene>.txt
TAG=${1:-r2}; shift
SCENES=${@:-cornell}
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
[ -f $OUT/counters_available.txt ] || rocprofv3 -L > $OUT/counters_available.txt 2>&1
for SCENE in $SCENES; do
  CMD="python3 $ROOT/bench.py --scene $SCENE --steps 24 --warmup 12 --no-cpu-baseline --frames-in-flight 1"
  P=0
  for SET in \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" \
    "SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" \
    "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_CYCLES SQ_THREAD_CYCLES_VALU" \
    "GRBM_GUI_ACTIVE GRBM_COUNT"
  do
    P=$((P+1))
    rm -rf $OUT/pipes_${SCENE}_$P
    rocprofv3 --pmc $SET --output-format csv -d $OUT/pipes_${SCENE}_$P -o pmc -- $CMD > $OUT/pipes_${SCENE}_$P.log 2>&1
  done
  python3 - "$OUT" "$SCENE" <<'EOF'
import sys, glob, csv, collections
out, scene = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob("%s/pipes_%s_*/**/*counter_collection.csv" % (out, scene), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_standardRenderer<0," in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("%s/pipes_%s.txt" % (out, scene), "w") as o:
    for k in sorted(agg):
        o.write("%-26s %16.0f   (mean of %d launches)\n" % (k, sum(agg[k]) / len(agg[k]), len(agg[k])))
print(open("%s/pipes_%s.txt" % (out, scene)).read())
EOF
done
