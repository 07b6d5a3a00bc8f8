#!/bin/bash
# round 4: candidate order of the refinement pass on bit strips (SBM_LOCAL_ORDER: 0 = slots per frame, 2 = one frame-major list)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_lborder_$1; rm -rf $O; mkdir -p $O; cd $R
for o in 0 2 0 2; do
for cfg in "case1 --steps 300 --inflight 1 --frame tiled" "case1 --steps 500 --frame tiled" "case1 --steps 1000" "case1 --steps 300 --inflight 1"; do
  echo "== order=$o $cfg"
  SBM_LOCAL_ORDER=$o timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err.log > $O/bench.json || { tail -5 $O/err.log; continue; }
  python tools/kshow.py $O/bench.json > $O/k.txt; head -2 $O/k.txt
done
done
