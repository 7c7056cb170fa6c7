cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -4
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "scorenet or train_step or pair" 2>&1 | tail -4
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused bn2   ms/step', d['ms_per_step'], d['final_loss'])"
P3_BN2_FUSED=0 python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two launch  ms/step', d['ms_per_step'], d['final_loss'])"
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused bn2   ms/step', d['ms_per_step'], d['final_loss'])"
