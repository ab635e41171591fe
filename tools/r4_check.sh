#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_baseline_sizes.py -q -m gpu -s 2>&1 | grep -E "pixels_outside|passed|failed|mis-rounded" | cut -c1-400
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_baseline_sizes.py 2>&1 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
