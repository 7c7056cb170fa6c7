# r05: full GPU suite at the final commit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 2000 python -m pytest tests/ -q -m gpu > gpurun_out/r05/g43_suite.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r05/g43_suite.txt
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r05/g43_suite.txt | head -20
