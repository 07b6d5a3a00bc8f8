#!/bin/bash
# config 5's frames in smaller calls with several in flight (us per frame is what compares)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_c5p; rm -rf $O; mkdir -p $O
A="--no-cpu-baseline --config c5 --steps 40"
python bench.py $A > $O/b64_default.json 2>$O/err.log
for b in 8 12 24 32; do for n in 3 4; do
  python bench.py $A --batch $b --inflight $n > $O/b${b}_n$n.json 2>>$O/err.log
done; done
tail -2 $O/err.log
python tools/kshow.py $O/*.json | grep value
