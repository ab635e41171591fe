#!/bin/bash
# Development aid: the three BASELINE scenes, pipelined and one frame at a time, one line each (no CPU baseline).
# usage: tools/quick_bench.sh [label] ; extra environment (SOLR_HIP_*) is passed through
label=${1:-now}
for scene in cornell height_field molecule; do
  python bench.py --no-cpu-baseline --scene $scene --steps 200 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['config']['rates_note']
print('$label %-12s pipelined %.4f ms  kernel %.4f ms (%s)  one-at-a-time %s' % ('$scene', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_spread'], r['one_frame_at_a_time'].split(':')[-1].strip()))"
done
