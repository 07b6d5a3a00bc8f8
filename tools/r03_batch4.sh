#!/bin/bash
# frames per call with four batches in flight
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_b4; rm -rf $O; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
for b in 8 12 16 20 24 32; do python bench.py $A --batch $b > $O/batch$b.json 2>$O/err.log; done
python tools/kshow.py $O/batch8.json $O/batch12.json $O/batch16.json $O/batch20.json $O/batch24.json $O/batch32.json | grep value
