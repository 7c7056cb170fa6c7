#!/bin/bash
# r06 g42: half-wave LayerNorm forward (planes): tests, then the step same-box with P3_LN_HALF=0 / 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g42.txt
: > $O
timeout 1500 python -m pytest tests/test_x3_gpu.py tests/test_model_gpu.py -q -m gpu -x 2>&1 | tail -6 | tee -a $O
for i in 1 2 3; do
  for P in 0 1; do
    echo -n "ln_half=$P " >> $O
    P3_LN_HALF=$P timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
tail -7 $O
