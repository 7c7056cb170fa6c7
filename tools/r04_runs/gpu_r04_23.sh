cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_x3 -o st -- python bench.py --lean --precision fp32x3 --steps 6 --warmup 2 > gpurun_out/r04/x3_run.log 2>&1
find /tmp/pf_x3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04/r04_fp32x3_step_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r04/r04_fp32x3_step_kernel_stats.csv 9 32 > gpurun_out/r04/r04_fp32x3_step_summary.txt
cat gpurun_out/r04/r04_fp32x3_step_summary.txt
