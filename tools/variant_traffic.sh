#!/bin/bash
# Development aid: time + HBM-side traffic of the renderer for alternative builds (build/var/libsolr_hip_<tag>.so)
TAGS=$1; SCENE=${2:-cornell}; ROOT=$PWD; export TMPDIR=/tmp
for t in $TAGS; do
  lib=$ROOT/build/var/libsolr_hip_$t.so; [ "$t" = base ] && lib=$ROOT/sol-r_amd/csrc/libsolr_hip.so
  export SOLR_HIP_LIB=$lib
  ms=$(python bench.py --scene $SCENE --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])")
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/vt_$c; (cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/vt_$c -o pmc -- python3 $ROOT/bench.py --scene $SCENE --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
  done
  f=$(python3 tools/pmc_summary.py "/tmp/vt_FETCH_SIZE/*counter_collection.csv" "k_standardRenderer<0," | head -1 | awk '{print $2}')
  w=$(python3 tools/pmc_summary.py "/tmp/vt_WRITE_SIZE/*counter_collection.csv" "k_standardRenderer<0," | head -1 | awk '{print $2}')
  echo "$t $SCENE kernel_ms=$ms FETCH_KB=$f WRITE_KB=$w"
done
