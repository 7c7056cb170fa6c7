#!/bin/bash
# r06 g19: same-box A/B of the train step: wide weight-gradient tile on / off (P3_TN_WIDE), interleaved; then rocprofv3 kernel stats of the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_ab_tn.txt
: > $O
for i in 1 2 3; do
  for v in 1 0; do
    echo -n "P3_TN_WIDE=$v " >> $O
    P3_TN_WIDE=$v timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
cat $O
rm -rf /tmp/pf_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_stats -o st -- python bench.py --lean --steps 10 > gpurun_out/stats_run.log 2>&1
find /tmp/pf_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_fp32x3_step_kernel_stats_mid.csv \;
python tools/kstats.py gpurun_out/r06_fp32x3_step_kernel_stats_mid.csv 15 40 > gpurun_out/r06_fp32x3_step_summary_mid.txt
head -42 gpurun_out/r06_fp32x3_step_summary_mid.txt | cut -c1-150
