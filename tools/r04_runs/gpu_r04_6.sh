cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "gemm_tn or sinkhorn" 2>&1 | tail -3
python tools/mb_tn.py 2>&1 | grep slabs
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['final_loss'])"
P3_SIDE_SN=1 python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SN side: ms/step', d['ms_per_step'], d['final_loss'])"
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['final_loss'])"
P3_SIDE_SN=1 python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SN side: ms/step', d['ms_per_step'], d['final_loss'])"
