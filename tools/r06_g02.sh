#!/bin/bash
# r06 g02: A-stationary kernel: LDS-form epilogue check, then the ablation variants (AS only)
mkdir -p gpurun_out
O=gpurun_out/mb_as_2.txt
: > $O
P3_AS_VAR=1 timeout 300 python tools/mb_as.py check >> $O 2>&1
for v in 0 1 2 4 6 8 10; do
  P3_AS_VAR=$v timeout 200 python tools/mb_as.py as >> $O 2>&1
done
tail -120 $O
