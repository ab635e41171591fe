#!/bin/bash
# Development aid (round 4): the new tests and the three BASELINE scenes through bench.py, one summary line each.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_walk_bound_gpu.py tests/test_engine_probes_gpu.py -x -q -m gpu 2>&1 | tail -30
for s in cornell height_field molecule; do
  timeout 400 python bench.py --scene $s --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/bench_wb_$s.json 2> gpurun_out/bench_wb_$s.err
  echo $s rc $?
  python - "$s" <<'PY'
import json, sys
s = sys.argv[1]
d = json.loads(open("gpurun_out/bench_wb_%s.json" % s).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["step_ms_spread"], "kernel", d["roofline"]["kernel_ms"])
print("walk bound", d["roofline"].get("walk_bound_mrays"), d["roofline"].get("walk_bound"))
print(d["config"]["rates_note"])
PY
done
