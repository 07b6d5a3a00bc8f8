#!/bin/bash
# rows per work item of the streaming gradient kernel (level 0, level 1) vs the pipelined step (three batches in flight):
# fewer, longer items do less warm-up work in total but leave SIMDs under-subscribed within one launch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_hs; mkdir -p $O
for hs in 0 24,10 32,18 38,32 46,32 46,46 60,32 74,32; do
  SBM_QS_HS=$hs timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra-frames --no-strong-estimate > $O/hs$hs.json 2>$O/err.log || tail -3 $O/err.log
done
python tools/kshow.py $O/hs*,*.json $O/hs0.json | grep -v "roofline\|other frames"
