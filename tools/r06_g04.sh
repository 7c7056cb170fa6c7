#!/bin/bash
# r06 g04: A-stationary kernel v2 (loads-first epilogue, DMA behind the MFMAs, six-slot ring, bias through LDS): check, timing against the tile kernels, stage sums
mkdir -p gpurun_out
O=gpurun_out/mb_as_4.txt
: > $O
timeout 300 python tools/mb_as.py check >> $O 2>&1
timeout 300 python tools/mb_as.py time >> $O 2>&1
P3_AS_VAR=2 timeout 200 python tools/mb_as.py as >> $O 2>&1
echo "== P3_AS_VAR=1" >> $O; P3_AS_VAR=1 timeout 200 python tools/mb_as.py dbg >> $O 2>&1
grep -v amdgpu.ids $O | tail -80
