#!/bin/bash
# r06 g11: A-stationary kernel v7 (epilogue in two halves: starter first, finisher last; w_lo fragments single-buffered): check, timing, ablations
mkdir -p gpurun_out
O=gpurun_out/mb_as_11.txt
: > $O
timeout 300 python tools/mb_as.py check >> $O 2>&1
timeout 300 python tools/mb_as.py time >> $O 2>&1
for v in 2 3; do P3_AS_VAR=$v timeout 200 python tools/mb_as.py as >> $O 2>&1; done
grep -v amdgpu.ids $O | tail -48
