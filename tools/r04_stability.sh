#!/bin/bash
# round 4: the default line (1000 steps) and the driver's call (20 steps, 5 warm-up) in fresh processes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_stability; rm -rf $O; mkdir -p $O; cd $R
for i in 1 2 3 4 5 6 7 8; do
  python3 bench.py --no-cpu-baseline --no-extra-frames --no-strong-estimate > $O/long_$i.json 2> $O/long_$i.err
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-frames --no-strong-estimate > $O/k20_$i.json 2> $O/k20_$i.err
  echo "run $i done"
done
python3 - <<'PY' | tee $O/summary.txt
import json, os, glob
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r04_stability/'
for kind in ('long', 'k20'):
    rows = []
    for f in sorted(glob.glob(root + kind + '_*.json')):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            rows.append((d['value'] / 1e6, d['ms_per_step'] * 1e3, d['config']['launch']['path'], d['config']['launch']['slots'], d['config']['launch'].get('probe_us_per_step')))
        except Exception as e:
            rows.append((0, 0, 'ERR ' + str(e), 0, None))
    print(kind, 'steps:', 'M templates*Mpx/s', [round(r[0], 2) for r in rows])
    print('   us/step', [round(r[1], 1) for r in rows])
    for r in rows:
        print('   ', r[2], r[3], r[4])
PY
