#!/bin/bash
# r06 g23: the 3-op split (attn_tile.h split8 / SplitStage, gemm.hip lds_put_split, gemm_tn.hip) against the 4-op one: step same-box, then the op / backward / model tests
mkdir -p gpurun_out
O=gpurun_out/r06_ab_split.txt
: > $O
for i in 1 2 3; do
  for L in tmp_ab/libp3hip_split3.so tmp_ab/libp3hip_split4.so; do
    echo -n "$(basename $L) " >> $O
    P3HIP_LIB=$L timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
cat $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py tests/test_train_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee -a $O
