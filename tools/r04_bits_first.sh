#!/bin/bash
# round 4: first run of the bit-plane coarse pass -- parity tests, then bits against bytes on c3 / c4 / case1
set -e
mkdir -p gpurun_out/r04_1
timeout -k 10 900 python -m pytest tests/test_gpu_coarse_pruning.py tests/test_gpu_match.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r04_1/tests.log 2>&1 || { tail -30 gpurun_out/r04_1/tests.log; exit 1; }
tail -3 gpurun_out/r04_1/tests.log
i=0
for mode in bytes bits; do
  for cfg in "c3" "c4 --templates 4500" "case1 --steps 200 --inflight 1" "case1 --steps 200"; do
    i=$((i+1))
    echo "== $mode $cfg"
    SBM_COARSE=$mode timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>gpurun_out/r04_1/err_${i}.log > gpurun_out/r04_1/bench_${i}.json || { tail -5 gpurun_out/r04_1/err_${i}.log; continue; }
    python tools/kshow.py gpurun_out/r04_1/bench_${i}.json
  done
done
