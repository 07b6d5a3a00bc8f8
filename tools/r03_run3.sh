#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_3; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_host_batch.py tests/test_gpu_facade.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
bash tools/facade_latency.sh > $O/latency.log 2>&1; echo "latency rc=$?"; grep -E "latency|batch_pinned|GB/s|us" $O/latency.log | tail -12
