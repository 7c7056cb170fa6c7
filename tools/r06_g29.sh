#!/bin/bash
# r06 g29: (1) the paired S / dP products of the backward kernels against the LDS-DMA library; (2) the forward kernel with ONE workgroup per CU (padded LDS): do the two
# workgroups of a CU overlap at all?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g29.txt
: > $O
for i in 1 2; do
for L in tmp_ab/libp3hip_dma.so tmp_ab/libp3hip_pair.so; do
  echo -n "$L " >> $O
  P3HIP_LIB=$L timeout 300 python tools/mb_attn_x3.py 2>&1 | tail -3 | tr "\n" " " >> $O; echo >> $O
done; done
for PAD in 0 40000; do
  echo -n "pad_lds=$PAD " >> $O
  P3_ATTN_PAD_LDS=$PAD P3HIP_LIB=tmp_ab/libp3hip_pair.so timeout 300 python tools/mb_attn_x3.py 2>&1 | tail -3 | tr "\n" " " >> $O; echo >> $O
done
cat $O
