#!/bin/bash
# kernel timeline of the default (four batches in flight, stream launches so that the tracer sees every kernel) bench: who overlaps whom, where the GPU idles
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_tl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SBM_BENCH_NO_ADAPT=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline --no-extra-frames --no-strong-estimate --steps 300 --warmup 20 > $O/bench.json 2> $O/err.log || tail -5 $O/err.log
cd $R
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_stats.py $F > $O/timeline.txt 2>&1
cat $O/timeline.txt
