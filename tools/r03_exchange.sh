#!/bin/bash
# the exchange step on one GPU (a 1-rank communicator): the library's ncclAllGather and the torch.distributed fallback
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_exch; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
python bench.py $A --force-collective > $O/native.json 2>$O/err.log
SBM_BENCH_TORCH_GATHER=1 python bench.py $A --force-collective > $O/torch.json 2>>$O/err.log
python bench.py $A > $O/plain.json 2>>$O/err.log
tail -5 $O/err.log
python tools/kshow.py $O/*.json | grep "value"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_exch/*.json')):
    d=json.load(open(f)); print(f, d['config'].get('exchange'), d['config']['launch'])
PY
