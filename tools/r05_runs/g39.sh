# r05: full GPU suite at the final commit, then the round's artefacts (tools/final_prof_r05.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r05/g39_suite.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r05/g39_suite.txt
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r05/g39_suite.txt | head -20
bash tools/final_prof_r05.sh > gpurun_out/final_prof_r05.log 2>&1
tail -25 gpurun_out/final_prof_r05.log | cut -c1-400
