#!/bin/bash
# r06 g46: the 3 x 3 convolutions' weights as planes (conv only), step same-box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g46.txt
: > $O
for i in 1 2 3; do
  for P in 0 1; do
    echo -n "conv_w_planes=$P " >> $O
    P3_CONV_W_PLANES=$P timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
cat $O
