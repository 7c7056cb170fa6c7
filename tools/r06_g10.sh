#!/bin/bash
# r06 g10: A-stationary kernel v6 (DMA pieces interleaved into the MFMA steps): check, timing, then without epilogue / without epilogue and weight stream
mkdir -p gpurun_out
O=gpurun_out/mb_as_10.txt
: > $O
timeout 300 python tools/mb_as.py check >> $O 2>&1
timeout 300 python tools/mb_as.py time >> $O 2>&1
for v in 2 3; do P3_AS_VAR=$v timeout 200 python tools/mb_as.py as >> $O 2>&1; done
grep -v amdgpu.ids $O | tail -48
