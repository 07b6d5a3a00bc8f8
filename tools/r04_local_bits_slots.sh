#!/bin/bash
# round 4: candidate slots per frame of the one-wave-per-candidate refinement kernel (SBM_LOCAL_GRID), candidate-heavy frames
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_lbslots_$1; rm -rf $O; mkdir -p $O; cd $R
for g in 256 512 1024 2048 4096; do
for cfg in "case1 --steps 300 --inflight 1 --frame tiled" "case1 --steps 500 --frame tiled" "case1 --steps 500"; do
  echo "== slots=$g $cfg"
  SBM_LOCAL_GRID=$g timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err.log > $O/bench.json || { tail -5 $O/err.log; continue; }
  python tools/kshow.py $O/bench.json > $O/k.txt; head -2 $O/k.txt
done
done
