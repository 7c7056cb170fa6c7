#!/bin/bash
# r06 g22: A-stationary kernel: finisher priority on (default) / off (P3_AS_VAR=6), twice each
mkdir -p gpurun_out
O=gpurun_out/mb_as_22.txt
: > $O
for v in 0 6 0 6; do P3_AS_VAR=$v timeout 200 python tools/mb_as.py as 2>&1 | grep -E "P3_AS_VAR|qkv|fc1|linear1" >> $O; done
cat $O
