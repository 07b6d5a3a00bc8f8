#!/bin/bash
# round 4: frames per call x calls in flight, us per FRAME (default 16 x 4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_shape; rm -rf $O; mkdir -p $O; cd $R
for shape in "16 4" "16 3" "16 5" "16 6" "8 4" "8 6" "12 4" "24 4" "32 4" "32 3" "24 3"; do
  set -- $shape
  python3 bench.py --steps 400 --batch $1 --inflight $2 --no-cpu-baseline --no-extra-frames --no-strong-estimate > $O/b.json 2> $O/b.err || { echo "shape=$shape failed"; tail -3 $O/b.err; continue; }
  python3 - "$1" "$2" <<'PY'
import json, os, sys
d = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r04_shape/b.json').read().strip().splitlines()[-1])
print(f"batch {sys.argv[1]:>3s} x inflight {sys.argv[2]}: {d['config']['us_per_frame']:6.2f} us/frame  {d['value']/1e6:6.2f} M   path {d['config']['launch']['path']} slots {d['config']['launch']['slots']}")
PY
done | tee $O/summary.txt
