#!/bin/bash
# r06 g35: weight planes by 16-byte loads + the bias-gradient column sums parked: tests, then the step same-box for each switch, kernel table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g35.txt
: > $O
timeout 1500 python -m pytest tests/test_x3_gpu.py tests/test_train_gpu.py tests/test_backward_gpu.py tests/test_ffl_gpu.py -q -m gpu -x 2>&1 | tail -8 | tee -a $O
for i in 1 2 3; do
  for V in "P3_W_PLANES=0 P3_COLSUM_PARK=0" "P3_W_PLANES=1 P3_COLSUM_PARK=0" "P3_W_PLANES=0 P3_COLSUM_PARK=1" "P3_W_PLANES=1 P3_COLSUM_PARK=1"; do
    echo -n "$V " >> $O
    env $V timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
tail -13 $O
rm -rf /tmp/pf_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_stats -o st -- python bench.py --lean --steps 10 > gpurun_out/stats_run.log 2>&1
find /tmp/pf_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_stats_g35.csv \;
python tools/kstats.py gpurun_out/r06_stats_g35.csv 15 40 | tee -a $O
timeout 600 python tools/aten_sites.py fp32x3 > gpurun_out/r06_aten_sites.txt 2>&1
tail -40 gpurun_out/r06_aten_sites.txt
