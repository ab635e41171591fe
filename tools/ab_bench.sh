#!/bin/bash
# Development aid: A/B of builds of the engine library on the same box, interleaved (boxes differ by a few %).
#   bash tools/ab_bench.sh "scene ..." ab/libsolr_hip_A.so ab/libsolr_hip_B.so [more libraries]
# prints, per scene and library, `ms_per_step` (frames delivered, median of 9 regions of 200 steps) and the kernel's
# own time over three rounds
SCENES=$1; shift
LIB=sol-r_amd/csrc/libsolr_hip.so
cp $LIB /tmp/libsolr_hip_keep.so
export SOLR_BENCH_REGIONS=9
one() { python bench.py --no-cpu-baseline --no-walk-bound --scene $1 --steps 200 --warmup 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %-12s delivered %.4f ms  kernel %.4f ms  one-at-a-time %s' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['rates_note']['one_frame_at_a_time'].split(':')[-1].strip()))"; }
for s in $SCENES; do
  for round in 1 2 3; do
    for l in "$@"; do
      cp $l $LIB; one $s $(basename $l)
    done
  done
done
cp /tmp/libsolr_hip_keep.so $LIB
