#!/bin/bash
# round 4: gradient-kernel change check: stream-kernel parity tests, gradient fuzz, then the default bench (1000 steps, twice) and the canvas frame
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_grad_$1; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_quantize_stream.py tests/test_gpu_stages.py tests/test_gpu_fuzz.py tests/test_gpu_match.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 600 python tools/fuzz_gradient.py 600 77 > $O/fuzz_gradient.log 2>&1 || { tail -20 $O/fuzz_gradient.log; exit 1; }
tail -1 $O/fuzz_gradient.log
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-strong-estimate > $O/bench_$i.json 2>$O/err_$i.log || { tail -5 $O/err_$i.log; continue; }
  python tools/kshow.py $O/bench_$i.json > $O/k.txt; head -3 $O/k.txt
done
