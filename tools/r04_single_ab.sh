#!/bin/bash
# round 4: one frame at a time (the latency shape): coarse pass on bit planes against the byte kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_single; rm -rf $O; mkdir -p $O; cd $R
i=0
for mode in bits bytes bits bytes; do
for fr in scene case1; do
  i=$((i+1))
  echo "== $mode $fr"
  SBM_COARSE=$mode timeout -k 10 300 python bench.py --config case1 --steps 300 --inflight 1 --batch 1 --frame $fr --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err_${i}.log > $O/bench_${i}.json || { tail -5 $O/err_${i}.log; continue; }
  python tools/kshow.py $O/bench_${i}.json > $O/k.txt; head -2 $O/k.txt
done
done
