#!/bin/bash
# round 4: refinement kernel change check: parity tests that exercise it, then the tiled frame, the headline and c5
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_local_$1; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_match.py tests/test_gpu_device_path.py tests/test_gpu_fuzz.py tests/test_gpu_coarse_bits.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
i=0
for cfg in "case1 --steps 300 --inflight 1 --frame tiled" "case1 --steps 1000" "case1 --steps 500 --frame tiled" "c5"; do
  i=$((i+1))
  echo "== $cfg"
  timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err_${i}.log > $O/bench_${i}.json || { tail -5 $O/err_${i}.log; continue; }
  python tools/kshow.py $O/bench_${i}.json > $O/k.txt; head -2 $O/k.txt
done
