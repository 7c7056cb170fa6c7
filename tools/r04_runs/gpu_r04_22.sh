cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_backward_gpu.py -q -m gpu -k "train_step_gradients_vs_oracle" -s 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tail -14
