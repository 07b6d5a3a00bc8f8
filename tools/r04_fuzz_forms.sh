#!/bin/bash
# round 4: the randomised match parity run with each form of the bit-plane coarse pass forced
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_fuzz_forms; rm -rf $O; mkdir -p $O; cd $R
SBM_COARSE=bits SBM_BITS_DW=2 SBM_BITS_BLOCK=0 timeout -k 10 500 python tools/fuzz_match.py 150 505 > $O/dw2.log 2>&1 || { tail -20 $O/dw2.log; exit 1; }
tail -1 $O/dw2.log
SBM_COARSE=bits SBM_BITS_BLOCK=1 timeout -k 10 500 python tools/fuzz_match.py 150 606 > $O/block.log 2>&1 || { tail -20 $O/block.log; exit 1; }
tail -1 $O/block.log
SBM_COARSE=bits SBM_BITS_DW=1 SBM_BITS_BLOCK=0 SBM_BITS_WIDE=0 timeout -k 10 500 python tools/fuzz_match.py 100 707 > $O/narrow.log 2>&1 || { tail -20 $O/narrow.log; exit 1; }
tail -1 $O/narrow.log
