#!/bin/bash
# r06 g24: column slices of one weight matrix park side by side (the fusion conv's nine shifted weight-gradient products) - step same-box, then the training tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_ab_park.txt
: > $O
for i in 1 2 3; do
  for L in tmp_ab/libp3hip_park.so tmp_ab/libp3hip_split3.so; do
    echo -n "$(basename $L) " >> $O
    P3HIP_LIB=$L timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
cat $O
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_ffl_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee -a $O
rm -rf /tmp/pf_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_stats -o st -- python bench.py --lean --steps 10 > gpurun_out/stats_run.log 2>&1
find /tmp/pf_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_stats_park.csv \;
python tools/kstats.py gpurun_out/r06_stats_park.csv 15 60 | grep -E "tn_reduce|tn_flush|total kernel" | tee -a $O
