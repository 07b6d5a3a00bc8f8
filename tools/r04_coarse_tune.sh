#!/bin/bash
# round 4: quick figures of the coarse pass on bit planes after a change (c3, c4 at one rank's share, case1 one batch at a time)
# + SQ / FETCH counters of the c4 launch
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_tune_$1; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_coarse_pruning.py tests/test_gpu_coarse_bits.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
i=0
for wide in 0 1; do
for cfg in "c3" "c4 --templates 4500" "case1 --steps 300 --inflight 1" "case1 --steps 1000" "case1 --steps 300 --inflight 1 --frame tiled" "case1 --steps 300 --inflight 1 --batch 1"; do
  i=$((i+1))
  echo "== wide=$wide $cfg"
  SBM_BITS_WIDE=$wide timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra-frames --no-strong-estimate 2>$O/err_${i}.log > $O/bench_${i}.json || { tail -5 $O/err_${i}.log; continue; }
  python tools/kshow.py $O/bench_${i}.json > $O/k.txt; head -2 $O/k.txt
done
done
if [ "$2" = "pmc" ]; then
cd /tmp && export TMPDIR=/tmp
A="--config c4 --templates 4500 --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sqa -o p -- python3 $R/bench.py $A > $O/sqa.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sqb -o p -- python3 $R/bench.py $A > $O/sqb.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/bench.py $A > $O/fetch.log 2>&1
cd $R
(python3 tools/pmc_summary.py $(find $O/sqa -name "*counter_collection.csv" | head -1); python3 tools/pmc_summary.py $(find $O/sqb -name "*counter_collection.csv" | head -1); python3 tools/pmc_summary.py $(find $O/fetch -name "*counter_collection.csv" | head -1)) > $O/pmc_c4.txt 2>&1
grep "coarse_bits" $O/pmc_c4.txt
cd /tmp
A="--config case1 --steps 40 --warmup 5 --no-cpu-baseline --no-extra-frames --no-strong-estimate --inflight 1"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sqa1 -o p -- python3 $R/bench.py $A > $O/sqa1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sqb1 -o p -- python3 $R/bench.py $A > $O/sqb1.log 2>&1
cd $R
(python3 tools/pmc_summary.py $(find $O/sqa1 -name "*counter_collection.csv" | head -1); python3 tools/pmc_summary.py $(find $O/sqb1 -name "*counter_collection.csv" | head -1)) > $O/pmc_case1.txt 2>&1
grep "sbm::" $O/pmc_case1.txt
fi
