#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_inf; mkdir -p $O
A="--no-cpu-baseline --no-strong-estimate --no-extra-frames"
python bench.py $A --inflight 3 > $O/b16_i3.json 2>$O/err.log
GPU_MAX_HW_QUEUES=8 python bench.py $A --inflight 4 > $O/b16_i4_q8.json 2>>$O/err.log
GPU_MAX_HW_QUEUES=8 python bench.py $A --inflight 5 > $O/b16_i5_q8.json 2>>$O/err.log
GPU_MAX_HW_QUEUES=8 python bench.py $A --inflight 3 > $O/b16_i3_q8.json 2>>$O/err.log
python bench.py $A --inflight 2 > $O/b16_i2.json 2>>$O/err.log
python bench.py $A --inflight 3 --batch 32 > $O/b32_i3.json 2>>$O/err.log
python bench.py $A --inflight 3 --batch 24 > $O/b24_i3.json 2>>$O/err.log
python bench.py $A --inflight 3 --batch 8 > $O/b8_i3.json 2>>$O/err.log
python tools/kshow.py $O/*.json | grep "value"
