#!/bin/bash
# Development aid: time the three bench scenes with alternative builds of libsolr_hip.so
# (build/var/libsolr_hip_<tag>.so).  usage: bash tools/variants.sh "tag tag ..." ["scene ..."]
TAGS=$1; SCENES=${2:-"cornell height_field molecule"}
for t in $TAGS; do
  lib=$PWD/build/var/libsolr_hip_$t.so; [ "$t" = base ] && lib=$PWD/sol-r_amd/csrc/libsolr_hip.so
  for s in $SCENES; do
    SOLR_HIP_LIB=$lib python bench.py --scene $s --steps 40 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', '$s', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
