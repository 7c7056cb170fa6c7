#!/bin/bash
# r06 g27: where a wave of the fp32x3 attention kernels spends its loop time (s_memtime per phase, timing build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P3HIP_LIB=tmp_ab/libp3hip_attntime.so timeout 300 python tools/mb_attn_phases.py > gpurun_out/r06_attn_phases.txt 2>&1
cat gpurun_out/r06_attn_phases.txt
