#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_x3_gpu.py -q -m gpu -x -k "parked or planes_too or weight_as_planes or conv3x3_gather" 2>&1 | tail -30 > gpurun_out/r06_g41.txt
cat gpurun_out/r06_g41.txt
