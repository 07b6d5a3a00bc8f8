#!/bin/bash
# with the wave priorities in place: batches in flight x hardware queues, three processes each
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_ap; rm -rf $O; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
for r in 1 2 3; do
  for n in 3 4 5; do
    python bench.py $A --inflight $n > $O/q4_inflight${n}_$r.json 2>$O/err.log
    GPU_MAX_HW_QUEUES=8 python bench.py $A --inflight $n > $O/q8_inflight${n}_$r.json 2>$O/err.log
  done
done
python tools/kshow.py $O/*.json | grep value
