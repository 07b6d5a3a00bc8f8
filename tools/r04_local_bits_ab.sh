#!/bin/bash
# round 4: refinement pass on bit strips (SBM_LOCAL_BITS, csrc/sbm_local_bits.h): parity, then the bench configurations with both forms
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_lbits_$1; rm -rf $O; mkdir -p $O; cd $R
if [ "$2" != "notests" ]; then
timeout -k 10 1000 python -m pytest tests/test_gpu_refine_bits.py tests/test_gpu_match.py tests/test_gpu_device_path.py tests/test_gpu_fuzz.py tests/test_gpu_coarse_bits.py tests/test_gpu_coarse_pruning.py tests/test_gpu_configs.py tests/test_gpu_host_batch.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 600 python tools/fuzz_match.py 200 909 > $O/fuzz.log 2>&1 || { tail -20 $O/fuzz.log; exit 1; }
tail -1 $O/fuzz.log
fi
i=0
for v in 0 1 0 1; do
for cfg in "case1 --steps 300 --inflight 1 --frame tiled" "case1 --steps 1000" "case1 --steps 300 --inflight 1" "case1 --steps 500 --frame tiled" "c5" "case1 --steps 300 --inflight 1 --batch 1"; do
  i=$((i+1))
  echo "== local_bits=$v $cfg"
  SBM_LOCAL_BITS=$v timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err_${i}.log > $O/bench_${i}.json || { tail -5 $O/err_${i}.log; continue; }
  python tools/kshow.py $O/bench_${i}.json > $O/k.txt; head -2 $O/k.txt
done
done
