#!/bin/bash
# round 4: the reference-shaped call Detector::match(cv::Mat) from host memory: upload in row bands (SBM_MATCH_BANDS) against one copy
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_latency; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_match.py tests/test_gpu_facade.py tests/test_gpu_stages.py tests/test_gpu_coarse_bits.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()"
for nb in 0 4 0 4 8 2; do
  echo "== SBM_MATCH_BANDS=$nb"
  SBM_MATCH_BANDS=$nb bash tools/facade_latency.sh 2>&1 | grep -v "^\[" | head -12
done
