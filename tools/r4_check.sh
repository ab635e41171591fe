#!/bin/bash
timeout 1500 python -m pytest tests/test_bench_launcher.py tests/test_in_process_devices_gpu.py -x -q -m gpu 2>&1 | tail -15
