#!/bin/bash
# round 4: coarse pass on bit planes, 32 against 64 positions per lane (SBM_BITS_DW)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_dw_$1; rm -rf $O; mkdir -p $O
cd $R
for dw in 1 2; do
SBM_BITS_DW=$dw timeout -k 10 600 python -m pytest tests/test_gpu_coarse_pruning.py tests/test_gpu_coarse_bits.py tests/test_gpu_match.py -x -q -m gpu > $O/tests_$dw.log 2>&1 || { tail -30 $O/tests_$dw.log; exit 1; }
tail -1 $O/tests_$dw.log
done
i=0
for dw in 1 2 0; do
for cfg in "case1 --steps 300 --inflight 1" "case1 --steps 1000" "case1 --steps 300 --inflight 1 --frame tiled" "c3" "c5" "case1 --steps 300 --inflight 1 --batch 1"; do
  i=$((i+1))
  echo "== dw=$dw $cfg"
  SBM_BITS_DW=$dw timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err_${i}.log > $O/bench_${i}.json || { tail -5 $O/err_${i}.log; continue; }
  python tools/kshow.py $O/bench_${i}.json > $O/k.txt; head -2 $O/k.txt
done
done
