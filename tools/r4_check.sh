#!/bin/bash
# experiment: how the delivered frame's copy is routed (copy stream or the frame's own stream) x buffer sets x host lag
export SOLR_BENCH_REGIONS=9
one() { python bench.py --no-cpu-baseline --no-walk-bound --scene cornell --steps 200 --warmup 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 delivered %.4f ms (%.4f .. %.4f)' % (d['ms_per_step'], d['config']['step_ms_spread']['min'], d['config']['step_ms_spread']['max']))"; }
for round in 1 2; do
for inline in 0 1; do for sets in 1 2 3 4; do for lag in 2 3; do
  SOLR_HIP_COPY_INLINE=$inline SOLR_BENCH_ENGINE_SETS=$sets SOLR_BENCH_LAG=$lag one "inline=$inline sets=$sets lag=$lag"
done; done; done; done
