cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for i in 1 2; do
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default (atomics in bf16) ms/step', d['ms_per_step'], d['final_loss'])"
P3_DETERMINISTIC=2 python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P3_DETERMINISTIC=2         ms/step', d['ms_per_step'], d['final_loss'])"
done
P3_FORCE_COLLECTIVES=1 python bench.py --lean --steps 20 2>&1 | tail -1 > gpurun_out/r04/bench_rccl_single_rank.json
python -c "import json; d=json.load(open('gpurun_out/r04/bench_rccl_single_rank.json')); print('forced collectives ms/step', d['ms_per_step'], d.get('collectives'))"
