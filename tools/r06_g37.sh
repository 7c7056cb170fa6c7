#!/bin/bash
# r06 g37: attention output as planes + per-call tile field: tests, step, kernel table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g37.txt
: > $O
timeout 1500 python -m pytest tests/test_x3_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee -a $O
for i in 1 2 3; do
  timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
done
rm -rf /tmp/pf_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_stats -o st -- python bench.py --lean --steps 10 > gpurun_out/stats_run.log 2>&1
find /tmp/pf_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_stats_g37.csv \;
python tools/kstats.py gpurun_out/r06_stats_g37.csv 15 48 | tee -a $O
