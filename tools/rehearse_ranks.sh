#!/bin/bash
# Development aid: bench.py --gpus 8 (cfg1, cfg4) with every rank on GPU 0 and tests/loopback_rccl.c as the transport -
# the launcher, balanced strips, the depth halo, the gather and the gathered-frame check of the N = 8 path on a one-GPU box.
#   bash tools/rehearse_ranks.sh [r4]     (from the repo root; results in gpurun_out/<tag>/rehearsal_8ranks_*.json)
# Three timed regions (SOLR_BENCH_REGIONS): the rates of a rehearsal mean nothing - one GPU, messages staged through files.
TAG=${1:-r4}
mkdir -p build gpurun_out/${TAG}
gcc -O2 -shared -fPIC -o build/libloopback_rccl.so tests/loopback_rccl.c -ldl || exit 1
for cfg in cfg1 cfg4; do
  D=$(mktemp -d -p /dev/shm solr_rehearsal_XXXX)
  extra=""; [ $cfg = cfg4 ] && extra="--config cfg4"
  SOLR_BENCH_SHARE_GPU=1 SOLR_HIP_RCCL_LIBRARY=$PWD/build/libloopback_rccl.so SOLR_LOOPBACK_DIR=$D SOLR_LOOPBACK_TIMEOUT=120 SOLR_BENCH_TIMEOUT=900 SOLR_BENCH_REGIONS=3 \
    timeout 1000 python bench.py --gpus 8 --no-cpu-baseline $extra > gpurun_out/${TAG}/rehearsal_8ranks_$cfg.json 2> gpurun_out/${TAG}/rehearsal_8ranks_$cfg.err
  echo "$cfg rc=$?"; tail -c 700 gpurun_out/${TAG}/rehearsal_8ranks_$cfg.json; echo; tail -3 gpurun_out/${TAG}/rehearsal_8ranks_$cfg.err
  rm -rf $D
done
