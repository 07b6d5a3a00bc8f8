set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, os, sys
sys.path.insert(0, '.')
import bench
from shape_based_matching_amd.templates import write_class_yaml
os.makedirs('gpurun_out/r2j', exist_ok=True)
ts = bench.case1_templates(360); ts.class_ids=["test"]
write_class_yaml(ts, 'gpurun_out/r2j/test_templ.yaml')
fr = bench.case1_frame("case1", 1024, 1024)
rgb = np.ascontiguousarray(fr[:, :, ::-1])
open('gpurun_out/r2j/frame.ppm','wb').write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]) + rgb.tobytes())
g = np.ascontiguousarray(fr[:, :, 1])
open('gpurun_out/r2j/frame.pgm','wb').write(b"P5\n%d %d\n255\n" % (g.shape[1], g.shape[0]) + g.tobytes())
PY
shape_based_matching_amd/sbm_facade_demo latency gpurun_out/r2j/%s_templ.yaml test gpurun_out/r2j/frame.ppm 90 128 300
shape_based_matching_amd/sbm_facade_demo latency gpurun_out/r2j/%s_templ.yaml test gpurun_out/r2j/frame.pgm 90 128 300
python tools/pcie_rate.py
