#!/bin/bash
# Development aid: an older tree in a worktree (git worktree add -f ab/wt/<commit> <commit>; make -C ab/wt/<commit>/sol-r_amd -j8), with its own
# bench.py, against libraries of this tree, interleaved:  SCENES="height_field molecule" bash tools/ab_against_a_worktree.sh <libraries>
export SOLR_BENCH_REGIONS=9
ROOT=$PWD
LIB=sol-r_amd/csrc/libsolr_hip.so
cp $LIB /tmp/keep.so
one() { ( cd $1; python bench.py --no-cpu-baseline --no-walk-bound --scene $2 --steps 200 --warmup 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s %-12s delivered %.4f ms  kernel %.4f ms  one-at-a-time %s' % ('$3', '$2', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['rates_note']['one_frame_at_a_time'].split(':')[-1].strip()))" ); }
for round in 1 2 3; do
  for s in $SCENES; do
    one $ROOT/ab/wt/${WORKTREE:-00f634e} $s r4_end
    for l in "$@"; do cp $l $LIB; one $ROOT $s $(basename $l); done
  done
done
cp /tmp/keep.so $LIB
