#!/bin/bash
# copy the judged records of tools/r04_profile.sh from gpurun_out/r04_prof/ (scratch) to profiles/ (tracked); $1 = label (a, b, ...)
set -e
S=gpurun_out/r04_prof; L=${1:-a}; P=profiles
cp $S/bench_with_counters.json $P/r04_${L}_bench.json
cp $S/bench_k20_with_counters.json $P/r04_${L}_bench_driver_call_20_steps.json
cp $S/bench_tiled.json $P/r04_${L}_tiled_bench.json
cp $S/other_configs.jsonl $P/r04_${L}_other_configs.jsonl
cp $S/r04_pmc.json $P/r04_pmc.json
for c in case1 c3 c4 c5; do
  cp $S/${c}_kernel_stats.csv $P/r04_${L}_${c}_kernel_stats.csv
  cp $S/${c}_pmc_traffic.txt $P/r04_${L}_${c}_pmc_traffic.txt
  cp $S/${c}_pmc_sq.txt $P/r04_${L}_${c}_pmc_sq.txt
  cp $S/${c}_under_rocprof.json $P/r04_${L}_${c}_bench_under_rocprof.json
done
ls -la $P | grep r04_
