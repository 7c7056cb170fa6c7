cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "dual_operand" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "scorenet or train_step_gradients" 2>&1 | tail -4
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mask2 kernel ms/step', d['ms_per_step'], d['final_loss'])"
P3_MASK2_DW=0 python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gemm_tn      ms/step', d['ms_per_step'], d['final_loss'])"
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mask2 kernel ms/step', d['ms_per_step'], d['final_loss'])"
