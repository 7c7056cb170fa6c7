#!/bin/bash
# r06 g39: attention output as planes from the attention launch, step same-box against the to_planes pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g39.txt
: > $O
for i in 1 2 3; do
  for P in 0 1; do
    echo -n "attn_oplanes=$P " >> $O
    P3_ATTN_OPLANES=$P timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'], d.get('fwd_ms_per_batch'))" >> $O
  done
done
cat $O
