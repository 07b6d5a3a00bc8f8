#!/bin/bash
# usage (GPU box, repo root): tools/inflight_sweep.sh <out.txt>  -- repeatability of the step time across frames-per-call x slots
out=${1:-gpurun_out/inflight_sweep.txt}
: > $out
for cfg in "2 4" "4 4" "8 4" "16 4" "16 2" "16 3" "8 2"; do
  set -- $cfg
  for rep in 1 2 3; do
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-extra-frames --steps 400 --warmup 40 --batch $1 --inflight $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('batch $1 inflight $2 rep $rep  us/frame %.2f  one-slot us/frame %.2f' % (c['us_per_frame'], list(v for k,v in c.items() if k.startswith('ms_per_step_one'))[0]*1e3/c['frames_per_step']))" >> $out
  done
done
cat $out
