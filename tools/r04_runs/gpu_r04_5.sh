set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_stats -o st -- python bench.py --lean --steps 20 > gpurun_out/r04/stats_run.log 2>&1
find /tmp/pf_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04/r04_mid_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r04/r04_mid_kernel_stats.csv 25 70 > gpurun_out/r04/r04_mid_summary.txt
cat gpurun_out/r04/r04_mid_summary.txt
python -m pytest tests/test_backward_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "sinkhorn" 2>&1 | tail -4
python tools/mb_sinkhorn.py
