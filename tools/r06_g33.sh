#!/bin/bash
# r06 g33: the register-staged fp32x3 GEMM with the weight as planes (decoder, fusion conv): tests, then the step same-box with the switch off / on, kernel table of the on form
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g33.txt
: > $O
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_ffl_gpu.py tests/test_predict_demo_gpu.py -q -m gpu -x 2>&1 | tail -12 | tee -a $O
for i in 1 2 3; do
  for P in 0 1; do
    echo -n "w_planes=$P " >> $O
    P3_W_PLANES=$P timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
tail -7 $O
rm -rf /tmp/pf_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_stats -o st -- python bench.py --lean --steps 10 > gpurun_out/stats_run.log 2>&1
find /tmp/pf_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_stats_wpl.csv \;
python tools/kstats.py gpurun_out/r06_stats_wpl.csv 15 30 | tee -a $O
