#!/bin/bash
# fresh processes until the slow stream-launch mode shows: what the host spends inside the launch calls there
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_slow; rm -rf $O; mkdir -p $O
for r in $(seq 1 24); do
  python bench.py --no-cpu-baseline --no-strong-estimate --no-extra-frames --steps 200 > $O/run$r.json 2>$O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_slow/run*.json'), key=lambda x:int(x.split('run')[-1].split('.')[0])):
    d=json.load(open(f)); l=d['config']['launch']
    print(f.split('/')[-1], round(d['ms_per_step']*1e3,1), l['path'], l['probe_us_per_step']['stream launches'], l.get('probe_host_enqueue_us_per_step'))
PY
