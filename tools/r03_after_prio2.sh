#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_ap2; rm -rf $O; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
python bench.py $A --config c3 > $O/c3.json 2>$O/err.log
python bench.py $A --config c3 --inflight 1 > $O/c3_1.json 2>$O/err.log
python bench.py $A --config c3 --inflight 3 > $O/c3_3.json 2>$O/err.log
python bench.py $A --config c5 > $O/c5.json 2>$O/err.log
python bench.py $A > $O/scene.json 2>$O/err.log
python bench.py $A --frame tiled > $O/tiled.json 2>$O/err.log
python tools/kshow.py $O/*.json | grep value
