#!/bin/bash
# round 4: stress of the facade's lane pool -- 8 host threads x 200 calls on one Detector (4 lanes, then 8, then 1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_threads; rm -rf $O; mkdir -p $O; cd $R
python - <<'PY'
import numpy as np, os, sys
sys.path.insert(0, '.')
import bench
from shape_based_matching_amd.templates import write_class_yaml
ts = bench.case1_templates(360); ts.class_ids = ["test"]
write_class_yaml(ts, 'gpurun_out/r04_threads/test_templ.yaml')
img = np.load('tests/golden/case1_test_bgr.npz')['bgr']
rgb = np.ascontiguousarray(img[:, :, ::-1])
open('gpurun_out/r04_threads/test.ppm', 'wb').write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]) + rgb.tobytes())
PY
for lanes in 4 8 1; do
  SECONDS=0
  shape_based_matching_amd/sbm_facade_demo threads gpurun_out/r04_threads/%s_templ.yaml test gpurun_out/r04_threads/test.ppm 88 128 8 200 60 $lanes 2>&1 | tail -2; echo "lanes $lanes: $SECONDS s"
done
