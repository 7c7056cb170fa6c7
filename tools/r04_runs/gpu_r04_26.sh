cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backward_gpu.py -q -m gpu -k "train_step_gradients_vs_oracle_autograd and fp32x3" -s 2>&1 | grep -E "^\[|Error|assert|^E " | head -20
timeout 900 python -m pytest tests/test_backward_gpu.py -q -m gpu -k "replaying_the_masks and fp32x3" 2>&1 | grep -E "Error|assert|^E " | head -12
