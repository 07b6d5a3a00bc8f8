#!/bin/bash
# distinct device copies of the frames used round-robin by the steps: 1 = every step reads one buffer (it stays in the Infinity Cache)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_ring; rm -rf $O; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
for r in 1 4 8 12; do  # --input-ring
  python bench.py $A --input-ring $r > $O/ring$r.json 2>$O/err.log
  python bench.py $A --input-ring $r > $O/ring${r}_b.json 2>$O/err.log
done
tail -2 $O/err.log
python tools/kshow.py $O/*.json | grep value
