#!/bin/bash
# A/B of the coarse pass's two kernels (SBM_COARSE=block|wave) over the bench configurations; run on the GPU box.
# usage: tools/coarse_ab.sh <outdir>
out=${1:-gpurun_out/coarse_ab}
mkdir -p "$out"
for mode in block wave; do
  export SBM_COARSE=$mode
  python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $out/case1_$mode.json 2>>$out/err.log
  python bench.py --no-cpu-baseline --steps 300 --warmup 50 --frame tiled --no-extra-frames > $out/tiled_$mode.json 2>>$out/err.log
  python bench.py --no-cpu-baseline --steps 300 --warmup 50 --batch 1 --no-extra-frames > $out/single_$mode.json 2>>$out/err.log
  python bench.py --no-cpu-baseline --config c3 --steps 50 --warmup 5 > $out/c3_$mode.json 2>>$out/err.log
  python bench.py --no-cpu-baseline --config c4 --templates 720 --steps 10 --warmup 2 > $out/c4_$mode.json 2>>$out/err.log
  python bench.py --no-cpu-baseline --config c5 --steps 20 --warmup 3 > $out/c5_$mode.json 2>>$out/err.log
  echo "$mode done"
done
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    k = d.get("kernels", {})
    print(os.path.basename(f), "ms/step %.4f" % d["ms_per_step"], {n: v.get("launch_us") for n, v in k.items()} if isinstance(k, dict) else "")
PY
