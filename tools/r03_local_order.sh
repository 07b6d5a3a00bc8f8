#!/bin/bash
# refinement pass: which candidate a workgroup takes (SBM_LOCAL_ORDER 0 / 1 / 2, see k_similarity_local), on config 5,
# the tiled frame and the default bench frame; one batch at a time and pipelined
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_lo3; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
for o in 0 2; do
  export SBM_LOCAL_ORDER=$o
  python bench.py $A --config c5 --steps 20 > $O/c5_o$o.json 2>$O/err.log
  python bench.py $A --frame tiled --inflight 1 --steps 300 > $O/tiled_1_o$o.json 2>>$O/err.log
  python bench.py $A --frame tiled > $O/tiled_3_o$o.json 2>>$O/err.log
  python bench.py $A > $O/scene_3_o$o.json 2>>$O/err.log
  python bench.py $A --inflight 1 --steps 300 > $O/scene_1_o$o.json 2>>$O/err.log
done
tail -3 $O/err.log
python tools/kshow.py $O/*.json | grep "value\|kernels us"
