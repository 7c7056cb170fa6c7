cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_backward_gpu.py -q -m gpu -k "bf16x3" 2>&1 | tail -6
timeout 1200 python -m pytest tests/test_backward_gpu.py -q -m gpu -k "train_step_gradients_vs_oracle" -s 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tail -14
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "forward_eval_vs_oracle" -s 2>&1 | grep -E "logits|passed|failed" | tail -14
python bench.py --lean --precision fp32 --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32    ms/step', d['ms_per_step'], d['value'], d['final_loss'])"
python bench.py --lean --precision fp32x3 --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32x3  ms/step', d['ms_per_step'], d['value'], d['final_loss'])"
