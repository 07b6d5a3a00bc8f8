#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_lg; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
for g in 512 128 64 256 512 128; do
  SBM_LOCAL_GRID=$g python bench.py $A > $O/g${g}_$RANDOM.json 2>$O/err.log
  SBM_LOCAL_GRID=$g python bench.py $A --frame tiled > $O/t${g}_$RANDOM.json 2>$O/err.log
done
python tools/kshow.py $O/*.json | grep -v "roofline\|other frames"
