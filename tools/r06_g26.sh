#!/bin/bash
# r06 g26: forward attention with the S-phase fragment reads pipelined (two chains): microbenchmark against the previous library, attention tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g26.txt
: > $O
for i in 1 2; do
for L in tmp_ab/libp3hip_base25.so pixelspointspolygons_amd/libp3hip.so; do
  echo -n "$L " >> $O
  P3HIP_LIB=$L timeout 300 python tools/mb_attn_x3.py 2>&1 | tail -3 | tr "\n" " " >> $O; echo >> $O
done; done
cat $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py tests/test_x3_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee -a $O
