#!/bin/bash
# A/B of two builds of libsbm_hip.so in one session (alternating, three rounds): pipelined step and kernels alone
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_ab; mkdir -p $O
for r in 1 2 3; do for v in prev new; do
  cp tools/bin/libsbm_hip_$v.so shape_based_matching_amd/libsbm_hip.so
  python bench.py --no-cpu-baseline --no-strong-estimate --no-extra-frames > $O/${v}_$r.json 2>$O/err.log || tail -3 $O/err.log
  python bench.py --no-cpu-baseline --no-strong-estimate --no-extra-frames --frame tiled > $O/${v}_tiled_$r.json 2>$O/err.log || tail -3 $O/err.log
done; done
python tools/kshow.py $O/*.json | grep -v "roofline\|other frames"
