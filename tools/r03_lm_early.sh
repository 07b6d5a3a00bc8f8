#!/bin/bash
# linear memories of a level right behind its gradient launch (two launches per batch) against one launch at the end
# (the SBM_LM_EARLY knob existed for this measurement only: early 110.6 - 111.4, one launch 110.0 - 110.9 us per step)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_lme; rm -rf $O; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
for r in 1 2 3; do
  python bench.py $A > $O/late_$r.json 2>$O/err.log
  SBM_LM_EARLY=1 python bench.py $A > $O/early_$r.json 2>$O/err.log
  python bench.py $A --frame tiled > $O/late_tiled_$r.json 2>$O/err.log
  SBM_LM_EARLY=1 python bench.py $A --frame tiled > $O/early_tiled_$r.json 2>$O/err.log
done
python bench.py --no-cpu-baseline --config c5 --steps 20 > $O/late_c5.json 2>$O/err.log
SBM_LM_EARLY=1 python bench.py --no-cpu-baseline --config c5 --steps 20 > $O/early_c5.json 2>$O/err.log
python tools/kshow.py $O/*.json | grep value
