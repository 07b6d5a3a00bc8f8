#!/bin/bash
# round 4: the eight-waves-per-item latency form of the bit-plane coarse pass (SBM_BITS_BLOCK) -- parity, then one frame at a time
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_block; rm -rf $O; mkdir -p $O; cd $R
for v in 1 0; do
SBM_BITS_BLOCK=$v timeout -k 10 600 python -m pytest tests/test_gpu_coarse_pruning.py tests/test_gpu_coarse_bits.py tests/test_gpu_match.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests_$v.log 2>&1 || { tail -30 $O/tests_$v.log; exit 1; }
tail -1 $O/tests_$v.log
done
i=0
for v in 0 1 0 1; do
for fr in scene case1; do
  i=$((i+1))
  echo "== block=$v $fr"
  SBM_BITS_BLOCK=$v timeout -k 10 300 python bench.py --config case1 --steps 300 --inflight 1 --batch 1 --frame $fr --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err_${i}.log > $O/bench_${i}.json || { tail -5 $O/err_${i}.log; continue; }
  python tools/kshow.py $O/bench_${i}.json > $O/k.txt; head -2 $O/k.txt
done
done
