#!/bin/bash
# waves per SIMD vs rows per wave for the streaming gradient kernel (textured 16 x 1024^2 x 3 step), GPU box.
# SBM_QS_LDS pads the launch with unused dynamic LDS: > 53.4 KB -> 2 workgroups per CU, > 80 KB -> 1.
out=${1:-gpurun_out/qsocc}
mkdir -p $out
run() { # name hs lds
  SBM_QS_HS=$2 SBM_QS_LDS=$3 python bench.py --no-cpu-baseline --steps 200 --warmup 30 --frame tiled --no-extra-frames --inflight 1 > $out/$1.json 2>>$out/err.log
  python - $out/$1.json $1 <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], [round(x, 1) for x in d["kernels"]["k_quantize"]["launch_us"]], flush=True)
PY
}
run base 0 0
run hs42_w2 42 60000
run hs40_w2 40 60000
run hs36_w2 36 60000
run hs28_w2 28 60000
run hs16_w2 16 60000
run hs12_w2 12 60000
