cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backward_gpu.py -q -m gpu -k "fp32x3 or attention_backward" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "attention" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "forward_eval_vs_oracle and fp32x3" -s 2>&1 | grep -E "logits|passed|failed" | tail -6
python bench.py --lean --precision fp32x3 --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32x3  ms/step', d['ms_per_step'], d['value'], d['final_loss'])"
