#!/bin/bash
# r06 g28: fp32x3 attention with LDS-DMA staging (raw fp32 images, split at the tile switch) + the statistics prefetch of the dK / dV kernel: microbenchmark against
# the previous library, the attention / backward / x3 tests, then the step same-box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g28.txt
: > $O
for i in 1 2; do
for L in tmp_ab/libp3hip_base25.so pixelspointspolygons_amd/libp3hip.so; do
  echo -n "$L " >> $O
  P3HIP_LIB=$L timeout 300 python tools/mb_attn_x3.py 2>&1 | tail -3 | tr "\n" " " >> $O; echo >> $O
done; done
cat $O
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py tests/test_x3_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee -a $O
for i in 1 2 3; do
  for L in tmp_ab/libp3hip_base25.so pixelspointspolygons_amd/libp3hip.so; do
    echo -n "$(basename $L) " >> $O
    P3HIP_LIB=$L timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
tail -7 $O
