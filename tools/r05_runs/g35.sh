cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -k "layer1_in_one_launch" 2>&1 | tail -3
