#!/bin/bash
# round 4: linear-memory builder A/B (SBM_LM_ALLTY: all four ty of a grid row per thread at the T = 4 strip level)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_lm_$1; rm -rf $O; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_match.py tests/test_gpu_device_path.py tests/test_gpu_stages.py tests/test_gpu_coarse_bits.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
i=0
for v in 0 1 0 1; do
for cfg in "case1 --steps 300 --inflight 1" "case1 --steps 1000"; do
  i=$((i+1))
  echo "== allty=$v $cfg"
  SBM_LM_ALLTY=$v timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err_${i}.log > $O/bench_${i}.json || { tail -5 $O/err_${i}.log; continue; }
  python tools/kshow.py $O/bench_${i}.json > $O/k.txt; head -2 $O/k.txt
done
done
