#!/bin/bash
# round 4: SQ / L2 counters of the refinement pass on bit strips on 16 candidate-heavy (tiled) frames
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_lbpmc_$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--config case1 --frame tiled --steps 30 --warmup 3 --inflight 1 --no-cpu-baseline --no-extra-frames --no-strong-estimate"
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sqa -o p -- python3 $R/bench.py $A > $O/sqa.log 2>&1 || tail -3 $O/sqa.log
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sqb -o p -- python3 $R/bench.py $A > $O/sqb.log 2>&1 || tail -3 $O/sqb.log
timeout -k 10 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/tcc -o p -- python3 $R/bench.py $A > $O/tcc.log 2>&1 || tail -3 $O/tcc.log
f() { ls $O/$1/*counter_collection.csv 2>/dev/null | head -1; }
(python3 $R/tools/pmc_summary.py $(f sqa); python3 $R/tools/pmc_summary.py $(f sqb); python3 $R/tools/pmc_summary.py $(f tcc)) > $O/summary.txt 2>&1
grep -E "k_similarity_local|k_build_lm|kernel" $O/summary.txt | head -40
