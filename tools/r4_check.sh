#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_multi_rank_gpu.py tests/test_bench_launcher.py tests/test_native_gather.py -x -q -m gpu 2>&1 | grep -E "DEBUG|AssertionError|Error|passed|failed" | head -20
