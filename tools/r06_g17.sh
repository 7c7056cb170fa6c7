#!/bin/bash
# r06 g17: same-box A/B of the train step with / without the A-stationary kernel (3 runs each, interleaved), then plain vs write-through epilogue stores
mkdir -p gpurun_out
O=gpurun_out/r06_ab_as.txt
: > $O
for i in 1 2 3; do
  for v in 1 0; do
    echo -n "P3_X3_AS=$v " >> $O
    P3_X3_AS=$v timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
cat $O
bash tools/r06_g16.sh
