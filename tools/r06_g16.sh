#!/bin/bash
# r06 g16: A-stationary kernel: plain epilogue stores against write-through (sc1) ones (P3_AS_VAR=5), same box, twice each
mkdir -p gpurun_out
O=gpurun_out/mb_as_16.txt
: > $O
for v in 0 5 0 5; do P3_AS_VAR=$v timeout 200 python tools/mb_as.py as 2>&1 | grep -E "P3_AS_VAR|qkv|fc1|dX fc2|linear1" >> $O; done
cat $O
