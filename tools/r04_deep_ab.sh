#!/bin/bash
# round 4: coarse pass on bit planes with one counter ripple per 32 features (P >= 10): parity, then c4 / c3 / case1
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_deep; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_coarse_pruning.py tests/test_gpu_coarse_bits.py tests/test_gpu_configs.py tests/test_gpu_stages.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
i=0
for cfg in "c4 --templates 4500" "c4 --templates 36000 --steps 3 --warmup 1" "c3" "case1 --steps 1000" "c5"; do
  i=$((i+1))
  echo "== $cfg"
  timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err_${i}.log > $O/bench_${i}.json || { tail -5 $O/err_${i}.log; continue; }
  python tools/kshow.py $O/bench_${i}.json > $O/k.txt; head -2 $O/k.txt
done
