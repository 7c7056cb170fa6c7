# r05 lease 7: both software-pipelined tiles, per shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_x3_gpu.py -q 2>&1 | tail -3
timeout 300 python tools/mb_x3.py > gpurun_out/r05/g07_mb_x3.txt 2>&1
cat gpurun_out/r05/g07_mb_x3.txt
