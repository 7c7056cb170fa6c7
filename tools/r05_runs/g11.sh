cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_backward_gpu.py -x -q -k "pillar_stem_backward and 16-12000" 2>&1 | grep -E "^E  |assert|passed|failed" | head -12
bash tools/final_prof_r05.sh > gpurun_out/final_prof_r05.log 2>&1
tail -30 gpurun_out/final_prof_r05.log
