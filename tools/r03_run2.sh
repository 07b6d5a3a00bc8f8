#!/bin/bash
# round 3, second GPU pass: banded-path tests, default bench with the strong-scaling estimate, band rehearsals
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_2; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_device_path.py tests/test_gpu_quantize_stream.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-frames --bands 8 > $O/bands8.json 2> $O/bands8.err; echo "bands8 rc=$?"; tail -3 $O/bands8.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-frames --bands 2 --force-collective > $O/bands2_coll.json 2> $O/bands2.err; echo "bands2 rc=$?"; tail -3 $O/bands2.err
python - <<'PY'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03_2'
for f in ('bench.json','bands8.json','bands2_coll.json'):
    try:
        d=json.loads(open(O+'/'+f).read().strip().splitlines()[-1])
        print(f, round(d['value']/1e6,2), round(d['ms_per_step']*1e3,1), {k:[round(x,1) for x in v['launch_us']] for k,v in d['kernels'].items()}, d['config']['parallelism'])
        se=d['config'].get('strong_estimate')
        if se:
            print('t1',se['one_gpu_kernels_us_per_step'])
            for n in ('2','4','8'):
                e=se[n]; print(n,'frames',round(e['frames']['rank_kernels_us'],1),round(e['frames']['speedup'],2),'| templates',round(e['templates']['rank_kernels_us'],1),round(e['templates']['speedup'],2),'| bands grad',round(e['bands']['gradient_us_slowest_band'],1),'rest',round(e['bands']['other_kernels_us'],1),'xchg',round(e['bands']['all_gather_model_us'],1),'speedup exp/hid',round(e['bands']['speedup_exchange_exposed'],2),round(e['bands']['speedup_exchange_hidden'],2))
    except Exception as e:
        print(f,'ERR',e)
PY
