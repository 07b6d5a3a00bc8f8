#!/bin/bash
# round 4: fused bit-plane producer -- parity tests (whole GPU suite), then the default bench and the single-batch shape
set -e
mkdir -p gpurun_out/r04_2
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04_2/tests.log 2>&1 || { tail -40 gpurun_out/r04_2/tests.log; exit 1; }
tail -3 gpurun_out/r04_2/tests.log
i=0
for cfg in "case1 --steps 300 --inflight 1" "case1 --steps 1000" "case1 --steps 20 --warmup 5" "c5" "case1 --steps 300 --frame tiled"; do
  i=$((i+1))
  echo "== $cfg"
  timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>gpurun_out/r04_2/err_${i}.log > gpurun_out/r04_2/bench_${i}.json || { tail -5 gpurun_out/r04_2/err_${i}.log; continue; }
  python tools/kshow.py gpurun_out/r04_2/bench_${i}.json
done
