#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_x3_gpu.py -q -m gpu -x -k "planes_too or block_stack" 2>&1 | tail -40 > gpurun_out/r06_g38.txt
cat gpurun_out/r06_g38.txt
for L in tmp_ab/libp3hip_base25.so pixelspointspolygons_amd/libp3hip.so; do P3HIP_LIB=$L timeout 300 python tools/mb_attn_x3.py 2>&1 | head -1; done | tee -a gpurun_out/r06_g38.txt
