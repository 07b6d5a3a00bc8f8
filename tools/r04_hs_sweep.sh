#!/bin/bash
# round 4: rows per work item of the streaming gradient kernel with four batches in flight (graph replay), us per 16-frame step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_hs; rm -rf $O; mkdir -p $O; cd $R
for hs in "" "24,10" "32,32" "32,18" "38,32" "46,32" "46,46" "60,32" "60,60" "74,32"; do
  SBM_QS_HS=$hs python3 bench.py --steps 600 --no-cpu-baseline --no-extra-frames --no-strong-estimate > $O/b.json 2> $O/b.err || { echo "hs=$hs failed"; tail -3 $O/b.err; continue; }
  python3 - "$hs" <<'PY'
import json, os, sys
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r04_hs/b.json').read().strip().splitlines()[-1])
print(f"SBM_QS_HS={sys.argv[1] or '(default)':10s} {d['ms_per_step']*1e3:7.1f} us/step  {d['value']/1e6:6.2f} M   one batch at a time {d['config']['ms_per_step_one_batch_at_a_time']*1e3:6.1f}  path {d['config']['launch']['path']}")
PY
done | tee $O/summary.txt
