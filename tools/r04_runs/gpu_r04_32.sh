cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "train_step" 2>&1 | tail -4
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deferred reduces ms/step', d['ms_per_step'], d['final_loss'])"
P3_DEFER_REDUCE=0 python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('immediate        ms/step', d['ms_per_step'], d['final_loss'])"
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deferred reduces ms/step', d['ms_per_step'], d['final_loss'])"
