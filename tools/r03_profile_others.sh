#!/bin/bash
# rocprofv3 kernel statistics of the other BASELINE configurations (c3, c4 at one rank's share, c5)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_prof_others; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "c3" "c4 --templates 4500" "c5"; do
  name=$(echo $cfg | cut -d' ' -f1)
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o t -- python3 $R/bench.py --no-cpu-baseline --config $cfg > $O/$name.json 2> $O/$name.err || tail -3 $O/$name.err
  find $O/$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  head -6 $O/${name}_kernel_stats.csv | cut -c1-140
done
