#!/bin/bash
# the default bench line in fresh processes: value and the launch path the probe picked; stock runtime and 8 hardware queues
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_stab; rm -rf $O; mkdir -p $O
for r in 1 2 3 4 5 6 7 8; do
  python bench.py --no-cpu-baseline --no-strong-estimate --no-extra-frames > $O/q4_run$r.json 2>$O/err.log
  GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --no-strong-estimate --no-extra-frames > $O/q8_run$r.json 2>$O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_stab/*run*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value']/1e6,2),'M', round(d['ms_per_step']*1e3,1),'us', d['config']['launch'])
PY
