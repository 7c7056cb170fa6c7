cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "gemm" 2>&1 | tail -12
