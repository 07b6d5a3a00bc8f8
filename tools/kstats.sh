#!/bin/bash
# usage: [KS_ARGS="--frame tiled"] tools/kstats.sh <label>  — rocprofv3 kernel-trace of a short single-stream bench run; per (kernel, grid) mean durations
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ks_$1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$1 -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-frames --inflight 1 $KS_ARGS > $GRAFT_REPO_ROOT/gpurun_out/ks_$1.log 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/ks_$1/ks_kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    d[(r['Kernel_Name'].split('(')[0].replace('void ','')[:32], r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k, v in sorted(d.items()):
    if 'k_' in k[0]:
        v = sorted(v); print("$1", k[0], k[1], "n=%d median %.2f us  p10 %.2f  p90 %.2f" % (len(v), v[len(v)//2], v[len(v)//10], v[9*len(v)//10]))
PY
