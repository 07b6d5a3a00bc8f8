#!/bin/bash
# four slots: replay of the captured graphs against stream launches, fresh processes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_gvs; rm -rf $O; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
for r in 1 2 3 4 5 6; do
  SBM_GRAPH=1 python bench.py $A > $O/graph$r.json 2>$O/err.log
  SBM_BENCH_NO_ADAPT=1 python bench.py $A > $O/stream$r.json 2>$O/err.log
done
python tools/kshow.py $O/*.json | grep value
