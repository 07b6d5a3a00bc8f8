#!/bin/bash
# candidate slots per frame of the refinement launch (SBM_LOCAL_GRID) on the bench step: case1 and tiled frames, GPU box
out=${1:-gpurun_out/lgrid}
mkdir -p $out
for g in 512 256 192 128 96 64 48 32; do
  for f in case1 tiled; do
    SBM_LOCAL_GRID=$g python bench.py --no-cpu-baseline --steps 200 --warmup 30 --frame $f --no-extra-frames > $out/${f}_$g.json 2>>$out/err.log
    python - $out/${f}_$g.json $f $g <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], "ms/step %.4f" % d["ms_per_step"], "local", [round(x, 1) for x in d["kernels"]["k_similarity_local"]["launch_us"]], flush=True)
PY
  done
done
