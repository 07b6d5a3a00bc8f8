#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_b32; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
python bench.py $A > $O/b16_default.json 2>$O/err.log
for hs in 32,32 46,32 60,32 60,46 74,46; do
  SBM_QS_HS=$hs python bench.py $A --batch 32 --steps 500 > $O/b32_hs$hs.json 2>>$O/err.log
done
SBM_QS_HS=46,32 python bench.py $A --batch 24 --steps 600 > $O/b24_hs46,32.json 2>>$O/err.log
python tools/kshow.py $O/*.json | grep "value"
