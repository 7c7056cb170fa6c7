#!/bin/bash
# r06 g12: A-stationary kernel v8 (epilogue halves through a per-wave LDS image: row-major, coalesced loads and stores; five-slot ring): check, timing, ablations
mkdir -p gpurun_out
O=gpurun_out/mb_as_12.txt
: > $O
timeout 300 python tools/mb_as.py check >> $O 2>&1
timeout 300 python tools/mb_as.py time >> $O 2>&1
for v in 2 3; do P3_AS_VAR=$v timeout 200 python tools/mb_as.py as >> $O 2>&1; done
grep -v amdgpu.ids $O | tail -48
