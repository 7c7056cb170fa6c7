cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python -m pytest tests/test_syncbn_gpu.py tests/test_rccl_single_rank_gpu.py -x -q -m gpu 2>&1 | tail -4
