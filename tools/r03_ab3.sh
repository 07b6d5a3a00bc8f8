#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_ab3; rm -rf $O; mkdir -p $O
cp shape_based_matching_amd/libsbm_hip.so /tmp/keep.so
for r in 1 2 3; do for v in prev new v2; do
  cp tools/bin/libsbm_hip_$v.so shape_based_matching_amd/libsbm_hip.so
  python bench.py --no-cpu-baseline --no-strong-estimate --no-extra-frames > $O/${v}_$r.json 2>$O/err.log || tail -3 $O/err.log
  python bench.py --no-cpu-baseline --no-strong-estimate --no-extra-frames --frame tiled > $O/${v}_tiled_$r.json 2>$O/err.log || tail -3 $O/err.log
done; done
cp /tmp/keep.so shape_based_matching_amd/libsbm_hip.so
python tools/kshow.py $O/*.json | grep value
