#!/bin/bash
# round 4: coarsest level of non-fusable grids (W * H % 256 != 0) as one spread plane + k_pack_bitplanes_spread: parity, then config 5
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_spack_$1; rm -rf $O; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py tests/test_gpu_coarse_bits.py tests/test_gpu_match.py tests/test_gpu_device_path.py tests/test_gpu_refine_bits.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for v in 1 0 1 0; do
  echo "== fused_bits=$v c5"
  SBM_FUSED_BITS=$v timeout -k 10 300 python bench.py --config c5 --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err.log > $O/bench_$v.json || { tail -5 $O/err.log; continue; }
  python tools/kshow.py $O/bench_$v.json > $O/k.txt; head -2 $O/k.txt
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err.log > $O/bench_head.json; python tools/kshow.py $O/bench_head.json | head -2
