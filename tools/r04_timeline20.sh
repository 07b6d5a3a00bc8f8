#!/bin/bash
# kernel timeline of the driver's call (20 steps, 5 warm-up; stream launches so that the tracer sees every kernel)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_tl20; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SBM_BENCH_NO_ADAPT=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-extra-frames --no-strong-estimate --steps 20 --warmup 5 > $O/bench.json 2> $O/err.log || tail -5 $O/err.log
cd $R
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
cp $F $O/kernel_trace.csv
python3 tools/kshow.py $O/bench.json | head -2
