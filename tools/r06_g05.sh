#!/bin/bash
# r06 g05: A-stationary kernel v3 (DMA by waves 0..3 only, L2 prefetch touches for the epilogue stream and the next A slice): check, timing, stage sums
mkdir -p gpurun_out
O=gpurun_out/mb_as_5.txt
: > $O
timeout 300 python tools/mb_as.py check >> $O 2>&1
timeout 300 python tools/mb_as.py time >> $O 2>&1
P3_AS_VAR=2 timeout 200 python tools/mb_as.py as >> $O 2>&1
echo "== P3_AS_VAR=1" >> $O; P3_AS_VAR=1 timeout 200 python tools/mb_as.py dbg >> $O 2>&1
grep -v amdgpu.ids $O | tail -80
