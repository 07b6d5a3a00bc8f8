#!/bin/bash
# round 4: the judged records.  1. default bench (1000 steps) and the driver's call (20 steps); 2. rocprofv3 kernel trace of the
# same command one batch at a time; 3. FETCH / WRITE / SQ counter passes (each in a run of its own) of case1, c3, c4 (one
# rank's share), c5; 4. the other configurations' bench lines.  Results under gpurun_out/r04_prof/; tools/r04_collect.sh
# copies what is judged to profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_prof; rm -rf $O; mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2>> $O/bench.err
python3 bench.py --frame tiled --no-cpu-baseline --no-strong-estimate --no-extra-frames > $O/bench_tiled.json 2>> $O/bench.err
echo "benches done"
cd /tmp && export TMPDIR=/tmp
pmc() { # name, bench args
  name=$1; shift
  A="$* --no-cpu-baseline --no-extra-frames --no-strong-estimate --inflight 1"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_trace -o t -- python3 $R/bench.py $A > $O/${name}_under_rocprof.json 2> $O/${name}_trace.err || tail -3 $O/${name}_trace.err
  find $O/${name}_trace -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${name}_fetch -o p -- python3 $R/bench.py $A > $O/${name}_fetch.log 2>&1 || tail -3 $O/${name}_fetch.log
  timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${name}_write -o p -- python3 $R/bench.py $A > $O/${name}_write.log 2>&1 || tail -3 $O/${name}_write.log
  timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/${name}_sqa -o p -- python3 $R/bench.py $A > $O/${name}_sqa.log 2>&1 || tail -3 $O/${name}_sqa.log
  timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${name}_sqb -o p -- python3 $R/bench.py $A > $O/${name}_sqb.log 2>&1 || tail -3 $O/${name}_sqb.log
  timeout -k 10 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${name}_tcc -o p -- python3 $R/bench.py $A > $O/${name}_tcc.log 2>&1 || tail -3 $O/${name}_tcc.log
  f() { find $O/${name}_$1 -name "*counter_collection.csv" | head -1; }
  python3 $R/tools/r04_pmc_json.py $O/r04_pmc.json $name $(f fetch) $(f write) $(f sqa) $(f sqb) $(f tcc) > $O/${name}_pmc_summary.txt 2>&1
  (python3 $R/tools/pmc_summary.py $(f fetch); python3 $R/tools/pmc_summary.py $(f write); python3 $R/tools/pmc_summary.py $(f tcc)) > $O/${name}_pmc_traffic.txt 2>&1
  (python3 $R/tools/pmc_summary.py $(f sqa); python3 $R/tools/pmc_summary.py $(f sqb)) > $O/${name}_pmc_sq.txt 2>&1
  echo "pmc $name done"
}
pmc case1 --config case1 --steps 40 --warmup 5
pmc c3 --config c3 --steps 40 --warmup 5
pmc c4 --config c4 --templates 4500 --steps 3 --warmup 1
pmc c5 --config c5 --steps 5 --warmup 2
cd $R
cp $O/r04_pmc.json profiles/r04_pmc.json   # so that the configuration lines below carry the counter-based fractions
: > $O/other_configs.jsonl
python3 bench.py --no-cpu-baseline --config c3 >> $O/other_configs.jsonl 2>> $O/bench.err
python3 bench.py --no-cpu-baseline --config c4 --templates 4500 >> $O/other_configs.jsonl 2>> $O/bench.err
python3 bench.py --no-cpu-baseline --config c4 --templates 36000 --steps 3 --warmup 1 >> $O/other_configs.jsonl 2>> $O/bench.err
python3 bench.py --no-cpu-baseline --config c5 >> $O/other_configs.jsonl 2>> $O/bench.err
python3 bench.py > $O/bench_with_counters.json 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_k20_with_counters.json 2>> $O/bench.err
head -12 $O/case1_kernel_stats.csv | cut -c1-160
python3 tools/kshow.py $O/bench.json $O/bench_k20.json $O/bench_tiled.json
python3 - <<'PY'
import json,os
for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r04_prof/other_configs.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:60], round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms', {k:[round(x,1) for x in v['launch_us']] for k,v in d['kernels'].items()}, 'roofline', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('kernel','frac','valu_issue_frac','l2_read_GBps','work_rate_GBps')})
PY
