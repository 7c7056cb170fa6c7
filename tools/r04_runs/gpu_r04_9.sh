cd $GRAFT_REPO_ROOT
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "bn_sums or pair_bwd_fused or scorenet or train_step or gemm_tn" 2>&1 | tail -8
for i in 1 2 3; do
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sums-from-G ms/step', d['ms_per_step'], d['final_loss'])"
P3_SUMS_FROM_G=0 python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('row pass    ms/step', d['ms_per_step'], d['final_loss'])"
done
