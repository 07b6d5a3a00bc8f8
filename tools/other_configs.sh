#!/bin/bash
# the other BASELINE configurations on one GPU; GPU box.  usage: tools/other_configs.sh <outdir>
out=${1:-gpurun_out/oc}
mkdir -p $out
python bench.py --no-cpu-baseline --config c3 --steps 100 --warmup 10 > $out/c3.json 2>>$out/err.log
python bench.py --no-cpu-baseline --config c4 --templates 720 --steps 10 --warmup 2 > $out/c4_720.json 2>>$out/err.log
python bench.py --no-cpu-baseline --config c4 --templates 3600 --steps 4 --warmup 1 > $out/c4_3600.json 2>>$out/err.log
python bench.py --no-cpu-baseline --config c5 --steps 40 --warmup 5 > $out/c5.json 2>>$out/err.log
SBM_GRAPH=1 python bench.py --no-cpu-baseline --config c5 --steps 40 --warmup 5 > $out/c5_graph.json 2>>$out/err.log
python bench.py --no-cpu-baseline --steps 300 --warmup 50 --batch 1 --inflight 1 --no-extra-frames > $out/single_1x1.json 2>>$out/err.log
python bench.py --no-cpu-baseline --steps 300 --warmup 50 --frame stagea --no-extra-frames > $out/stagea.json 2>>$out/err.log
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    k = d.get("kernels", {})
    print(os.path.basename(f), "ms/step %.4f" % d["ms_per_step"], "value %.4g" % d["value"], {n: [round(x, 1) for x in v.get("launch_us")] for n, v in k.items()} if isinstance(k, dict) else "", flush=True)
PY
