#!/bin/bash
# round 3: the judged records.  1. default bench; 2. rocprofv3 kernel trace + FETCH / WRITE / SQ counter passes of the same
# command (one batch at a time: --inflight 1); 3. the other BASELINE configurations.  Results under gpurun_out/r03_prof/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_prof; rm -rf $O; mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 bench.py --frame tiled --no-cpu-baseline --no-strong-estimate --no-extra-frames > $O/bench_tiled.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
A="--no-cpu-baseline --no-extra-frames --no-strong-estimate --inflight 1"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py $A > $O/bench_under_rocprof.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/bench.py $A --steps 40 --warmup 5 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/bench.py $A --steps 40 --warmup 5 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sqa -o p -- python3 $R/bench.py $A --steps 40 --warmup 5 > $O/sqa.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sqb -o p -- python3 $R/bench.py $A --steps 40 --warmup 5 > $O/sqb.log 2>&1
cd $R
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python3 tools/make_pmc_json.py $F $W $O/pmc_traffic.json > /dev/null
python3 tools/pmc_summary.py $F > $O/pmc_fetch.txt; python3 tools/pmc_summary.py $W > $O/pmc_write.txt
(python3 tools/pmc_summary.py $(find $O/sqa -name "*counter_collection.csv" | head -1); python3 tools/pmc_summary.py $(find $O/sqb -name "*counter_collection.csv" | head -1)) > $O/pmc_sq.txt 2>&1
# other BASELINE configurations (one JSON line each)
: > $O/other_configs.jsonl
python3 bench.py --no-cpu-baseline --config c3 >> $O/other_configs.jsonl 2>> $O/bench.err
python3 bench.py --no-cpu-baseline --config c4 --templates 4500 >> $O/other_configs.jsonl 2>> $O/bench.err
python3 bench.py --no-cpu-baseline --config c4 --templates 36000 --steps 3 --warmup 1 >> $O/other_configs.jsonl 2>> $O/bench.err
python3 bench.py --no-cpu-baseline --config c5 >> $O/other_configs.jsonl 2>> $O/bench.err
head -12 $O/kernel_stats.csv | cut -c1-160
python3 tools/kshow.py $O/bench.json $O/bench_tiled.json
python3 - <<'PY'
import json,os
for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03_prof/other_configs.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:60], round(d['value']/1e6,2),'M', round(d['ms_per_step'],3),'ms', {k:[round(x,1) for x in v['launch_us']] for k,v in d['kernels'].items()})
PY
