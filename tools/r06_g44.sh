#!/bin/bash
# r06 g44: dX GEMMs of the per-operator fp32x3 path on the optimizer's transposed weight planes (no fp32 transposes derived): tests, step same-box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g44.txt
: > $O
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_model_gpu.py tests/test_backward_gpu.py -q -m gpu -x 2>&1 | tail -6 | tee -a $O
for i in 1 2 3; do
  for P in 0 1; do
    echo -n "wt_planes=$P " >> $O
    P3_WT_PLANES=$P timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
tail -7 $O
