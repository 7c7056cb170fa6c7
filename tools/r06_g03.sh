#!/bin/bash
# r06 g03: A-stationary kernel: s_memtime stage sums (VAR 64 / 66), then deep prefetch / two accumulators at K = 256 (VAR 16 / 32 / 48, +2 = no epilogue)
mkdir -p gpurun_out
O=gpurun_out/mb_as_3.txt
: > $O
for v in 64 66; do echo "== P3_AS_VAR=$v" >> $O; P3_AS_VAR=$v timeout 200 python tools/mb_as.py dbg >> $O 2>&1; done
for v in 0 16 32 48 2 18 34 50; do
  P3_AS_VAR=$v timeout 200 python tools/mb_as.py as 2>&1 | grep -E "P3_AS_VAR|dec " >> $O
done
grep -v amdgpu.ids $O | tail -80
