#!/bin/bash
# experiment: more hardware queues for the process (GPU_MAX_HW_QUEUES, ROCm's default is 4)
export SOLR_BENCH_REGIONS=9
one() { python bench.py --no-cpu-baseline --no-walk-bound --height $2 --steps 300 --warmup 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 rows $2: delivered %.4f ms (%.4f .. %.4f)' % (d['ms_per_step'], d['config']['step_ms_spread']['min'], d['config']['step_ms_spread']['max']))"; }
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  for h in 1080 136; do
    SOLR_BENCH_COPY_INLINE=0 SOLR_BENCH_ENGINE_SETS=2 SOLR_BENCH_LAG=2 one "queues $q: copy stream, 2 sets, lag 2;" $h
    SOLR_BENCH_COPY_INLINE=0 SOLR_BENCH_ENGINE_SETS=3 SOLR_BENCH_LAG=3 one "queues $q: copy stream, 3 sets, lag 3;" $h
    SOLR_BENCH_COPY_INLINE=0 SOLR_BENCH_ENGINE_SETS=4 SOLR_BENCH_LAG=4 one "queues $q: copy stream, 4 sets, lag 4;" $h
    SOLR_BENCH_COPY_INLINE=1 SOLR_BENCH_ENGINE_SETS=3 SOLR_BENCH_LAG=3 one "queues $q: own streams, 3 sets, lag 3;" $h
    SOLR_BENCH_COPY_INLINE=1 SOLR_BENCH_ENGINE_SETS=4 SOLR_BENCH_LAG=4 one "queues $q: own streams, 4 sets, lag 4;" $h
  done
done
