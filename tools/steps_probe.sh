#!/bin/bash
# Development aid: ms_per_step of bench.py for short timed regions (the driver's 20 steps / 5 warm-up among them),
# optionally A/B between two builds of the engine library:  bash tools/steps_probe.sh [A.so B.so] [scene]
LIB=sol-r_amd/csrc/libsolr_hip.so
one() { python bench.py --no-cpu-baseline --scene $4 --steps $1 --warmup $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3 $4 steps $1 warmup $2: %.4f ms' % d['ms_per_step'])"; }
if [ -n "$2" ]; then A=$1; B=$2; SCENE=${3:-cornell}; cp $LIB /tmp/libsolr_hip_keep.so; else A=""; SCENE=${1:-cornell}; fi
for r in 1 2 3; do
  for cfg in "20 5" "20 40" "200 5"; do
    if [ -n "$A" ]; then cp $A $LIB; one $cfg A $SCENE; cp $B $LIB; one $cfg B $SCENE; else one $cfg - $SCENE; fi
  done
done
[ -n "$A" ] && cp /tmp/libsolr_hip_keep.so $LIB
