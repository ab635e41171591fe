#!/bin/bash
# Development aid (round 5): the N > 1 job's mode sweep with ONE rank of the real RCCL on a strip of an eighth of the frame
# (136 rows) - what a rank of an eight-GPU job does per step - three times: ms per delivered frame of every combination.
# The send-to-self of the one-rank communicator stands for the gather's cost on the sending side.
for round in 1 2 3; do
  SOLR_BENCH_FORCE_DIST=1 SOLR_BENCH_STRIP_SETTINGS=1 SOLR_BENCH_SWEEP=1 SOLR_BENCH_REGIONS=9 python bench.py --height 136 --steps 200 --warmup 40 \
      --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['config']['mode_sweep']
for n in s:
    if n != 'headline_runs_on':
        print('%-52s %s' % (n, ('%.4f ms per delivered frame' % s[n]['ms_per_step']) if 'ms_per_step' in s[n] else s[n]))
print('headline on', s['headline_runs_on'], ': %.4f ms per delivered frame' % d['ms_per_step'])"
done
