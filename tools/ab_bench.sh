#!/bin/bash
# Development aid: A/B of two builds of the engine library on the same box, interleaved (boxes differ by a few %).
#   bash tools/ab_bench.sh ab/libsolr_hip_A.so ab/libsolr_hip_B.so [scene ...]
A=$1; B=$2; shift 2; SCENES=${@:-cornell}
LIB=sol-r_amd/csrc/libsolr_hip.so
cp $LIB /tmp/libsolr_hip_keep.so
one() { python bench.py --no-cpu-baseline --scene $1 --steps 200 --warmup 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s %s pipelined %.4f ms  kernel %.4f ms' % ('$2', '$1', d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for s in $SCENES; do
  for round in 1 2 3; do
    cp $A $LIB; one $s A
    cp $B $LIB; one $s B
  done
done
cp /tmp/libsolr_hip_keep.so $LIB
