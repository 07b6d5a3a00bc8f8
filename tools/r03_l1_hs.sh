#!/bin/bash
# level-1 gradient launch alone (one batch at a time) against rows per work item: VERDICT round 2 item 7
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_l1hs; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames --inflight 1 --steps 200"
for hs in 4 6 8 10 12 14 18; do
  SBM_QS_HS=24,$hs SBM_BENCH_LATENCY_SIZING=1 python bench.py $A --frame tiled > $O/hs$hs.json 2>$O/err.log
done
tail -2 $O/err.log
python tools/kshow.py $O/hs4.json $O/hs6.json $O/hs8.json $O/hs10.json $O/hs12.json $O/hs14.json $O/hs18.json | grep "value\|kernels us"
