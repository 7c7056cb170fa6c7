#!/bin/bash
# r06 g30: forward fp32x3 attention at head dim 64 with 32-key tiles (three workgroups per CU) against 64-key tiles, LDS-DMA staging both
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g30.txt
: > $O
for i in 1 2; do
for L in pixelspointspolygons_amd/libp3hip.so tmp_ab/libp3hip_kt32.so; do
  echo -n "$L " >> $O
  P3HIP_LIB=$L timeout 300 python tools/mb_attn_x3.py 2>&1 | tail -3 | head -1 >> $O
done; done
cat $O
