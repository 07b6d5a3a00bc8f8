#!/bin/bash
# quick look at the coarse pass on the bench workloads (GPU box).  usage: tools/coarse_quick.sh <outdir>
out=${1:-gpurun_out/cq}
mkdir -p "$out"
python bench.py --no-cpu-baseline --steps 300 --warmup 50 > $out/case1.json 2>>$out/err.log
python bench.py --no-cpu-baseline --steps 300 --warmup 50 --frame tiled --no-extra-frames > $out/tiled.json 2>>$out/err.log
python bench.py --no-cpu-baseline --config c3 --steps 50 --warmup 5 > $out/c3.json 2>>$out/err.log
python bench.py --no-cpu-baseline --config c5 --steps 20 --warmup 3 > $out/c5.json 2>>$out/err.log
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    k = d.get("kernels", {})
    print(os.path.basename(f), "ms/step %.4f" % d["ms_per_step"], {n: [round(x, 1) for x in v.get("launch_us")] for n, v in k.items()} if isinstance(k, dict) else "")
PY
