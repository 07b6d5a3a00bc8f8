#!/bin/bash
# usage: [KS_ARGS="--frame tiled"] tools/pmc_sq.sh <label> [kernel-substring]
# two rocprofv3 --pmc passes (SQ occupancy / stall buckets, instruction mix) over a short single-stream bench run
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
rm -rf $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/a -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-frames --inflight 1 $KS_ARGS > $O.a.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/b -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-frames --inflight 1 $KS_ARGS > $O.b.log 2>&1 &&
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $O/a/p_counter_collection.csv $2 && python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $O/b/p_counter_collection.csv $2
