#!/bin/bash
# usage: runbench.sh <label>   (run on GPU box from repo root)
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -1 gpurun_out/gpu_tests.log
for n in 1 2; do timeout -k 10 300 python bench.py --steps 600 --warmup 60 --no-cpu-baseline --inflight $n > gpurun_out/dbgx.log 2>&1; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/dbgx.log").read().strip().splitlines()[-1]); print("$1 inflight", $n, round(d["ms_per_step"]*1e3,1), d["config"]["matches_distinct"], {k:(round(v["avg_launch_us"],1), v["launches"]) for k,v in d["kernels"].items()})
except Exception as e: print("FAIL", open("gpurun_out/dbgx.log").read()[-300:])
PY
done
