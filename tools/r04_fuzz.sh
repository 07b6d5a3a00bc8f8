#!/bin/bash
# round 4: whole GPU suite, then the randomised parity runs (match: 300 cases incl. the bit-plane coarse pass and the graph
# default; gradient: 1000 cases)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_fuzz; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 900 python tools/fuzz_match.py 300 404 > $O/fuzz_match.log 2>&1 || { tail -20 $O/fuzz_match.log; exit 1; }
tail -3 $O/fuzz_match.log
