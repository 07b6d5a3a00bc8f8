#!/bin/bash
# round 3, first GPU pass: GPU test-suite, default bench, the other frames, config 4 at 4500 templates
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_1; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
timeout -k 10 300 python bench.py --config c4 --templates 4500 --no-cpu-baseline > $O/c4_4500.json 2> $O/c4.err; echo "c4 rc=$?"; tail -3 $O/c4.err
python - <<'PY'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03_1'
for f in ('bench.json','c4_4500.json'):
    try:
        d=json.loads(open(O+'/'+f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], {k:v['launch_us'] for k,v in d['kernels'].items()})
        c=d['config']
        print({k:c[k] for k in c if k.startswith('value_') or k.endswith('us_per_frame')})
        print(d.get('roofline',{}).get('frac'), d.get('cpu_baseline',{}).get('value'))
    except Exception as e:
        print(f,'ERR',e)
PY
