cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_syncbn_gpu.py -x -q -m gpu -k "two_ranks_with_sync_batchnorm_equal and not ffl" -s 2>&1 | tail -40
echo "exit: $?"
