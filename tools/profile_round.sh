#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <label> [bench args]
# 1. the bench itself; 2. rocprofv3 --kernel-trace --stats of the same command; 3. FETCH_SIZE and WRITE_SIZE PMC passes
# (separate, as the MI355X guide prescribes).  Everything lands under gpurun_out/prof_<label>/; copy the summaries you
# want judged into profiles/ (tools/make_pmc_json.py turns the two PMC CSVs into profiles/pmc_traffic.json).
L=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$L
rm -rf $O; mkdir -p $O
python3 $R/bench.py "$@" > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
# --inflight 1: one stream, so the tracer sees each kernel alone, like bench.py's own per-kernel pass
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline --inflight 1 "$@" > $O/bench_under_rocprof.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/bench.py --no-cpu-baseline --inflight 1 --steps 40 --warmup 5 "$@" > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/bench.py --no-cpu-baseline --inflight 1 --steps 40 --warmup 5 "$@" > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
ls -R $O | head -40
tail -n 1 $O/bench.json | cut -c1-400
