#!/bin/bash
# the driver's call (--steps 20 --warmup 5) with the probe in place: ties go to graph replay (default) or to stream launches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_k20; rm -rf $O; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames --steps 20 --warmup 5"
for r in 1 2 3 4 5 6; do
  python bench.py $A > $O/graph$r.json 2>$O/err.log
  SBM_BENCH_PREFER=stream python bench.py $A > $O/stream$r.json 2>$O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_k20/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], round(d["value"]/1e6,2), round(d["ms_per_step"]*1e3,1), d["config"]["launch"]["path"], d["config"]["launch"]["slots"])
PY
