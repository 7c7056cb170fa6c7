cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused   ms/step', d['ms_per_step'], d['final_loss'])"
P3_PAIR_FUSED=0 python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unfused ms/step', d['ms_per_step'], d['final_loss'])"
done
