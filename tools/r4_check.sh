#!/bin/bash
(time timeout 2800 python -m pytest tests -q -m gpu 2>&1 | tail -6) 2>&1
(time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_like.json 2> gpurun_out/bench_driver_like.err) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_driver_like.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["regions"], d["config"]["step_ms_spread"], d["roofline"]["kernel_ms"], d["roofline"]["walk_bound_mrays"], d["cpu_baseline"]["value"])
PY
