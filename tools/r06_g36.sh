#!/bin/bash
# r06 g40 (final pass): the whole GPU suite, then the round's final artefacts (tools/final_prof_r06.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06_gpu_suite_run7.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r06_gpu_suite_run7.txt
tail -5 gpurun_out/r06_gpu_suite_run7.txt
bash tools/final_prof_r06.sh > gpurun_out/final_prof.log 2>&1
tail -30 gpurun_out/final_prof.log | cut -c1-400
