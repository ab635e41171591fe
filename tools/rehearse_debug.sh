python - <<'PY'
import sys, os
sys.path.insert(0, "tests")
from test_multi_rank_gpu import build_loopback
print(build_loopback())
PY
mkdir -p /dev/shm/solr_x; export SOLR_BENCH_SHARE_GPU=1 SOLR_HIP_RCCL_LIBRARY=$PWD/build/libloopback_rccl.so SOLR_LOOPBACK_DIR=/dev/shm/solr_x SOLR_LOOPBACK_TIMEOUT=60 SOLR_BENCH_REGIONS=3 SOLR_HIP_COMM_PER_FLIGHT=1
python bench.py --gpus 2 --no-cpu-baseline --steps 12 --warmup 3 --width 640 --height 360 > gpurun_out/rehearse_pf.json 2> gpurun_out/rehearse_pf.err; echo rc $?; tail -c 3000 gpurun_out/rehearse_pf.err
