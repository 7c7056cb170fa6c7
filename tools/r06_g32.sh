#!/bin/bash
# r06 g32: the whole GPU suite on the tree with the LDS-DMA attention staging, the conv parking and the rest of the round's second half
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_gpu_suite_run3.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r06_gpu_suite_run3.txt
tail -5 gpurun_out/r06_gpu_suite_run3.txt
