#!/bin/bash
# r06 g47: the whole GPU suite at the round's last commit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_gpu_suite_run8.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r06_gpu_suite_run8.txt
tail -4 gpurun_out/r06_gpu_suite_run8.txt
