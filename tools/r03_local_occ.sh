#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_locc; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
python bench.py $A --config c5 --steps 20 > $O/c5.json 2>$O/err.log
python bench.py $A --frame tiled --inflight 1 --steps 300 > $O/tiled_1.json 2>>$O/err.log
python bench.py $A --frame tiled > $O/tiled_3.json 2>>$O/err.log
python bench.py $A > $O/scene_3.json 2>>$O/err.log
python bench.py $A --inflight 1 --steps 300 > $O/scene_1.json 2>>$O/err.log
tail -3 $O/err.log
python tools/kshow.py $O/*.json | grep "value\|kernels us"
