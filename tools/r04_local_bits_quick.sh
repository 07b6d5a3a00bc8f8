#!/bin/bash
# round 4: quick look at the refinement pass on bit strips (one form, the candidate-heavy and the headline frames)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_lbq_$1; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_refine_bits.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
i=0
for cfg in "case1 --steps 300 --inflight 1 --frame tiled" "case1 --steps 1000" "case1 --steps 500 --frame tiled" "c5" "case1 --steps 300 --inflight 1 --batch 1"; do
  i=$((i+1))
  echo "== $cfg"
  timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err_${i}.log > $O/bench_${i}.json || { tail -5 $O/err_${i}.log; continue; }
  python tools/kshow.py $O/bench_${i}.json > $O/k.txt; head -2 $O/k.txt
done
