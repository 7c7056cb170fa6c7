# r05: fp32-family BatchNorm-2 backward statistics of the pillar stem on the 8-channel form (pfn_bwd_l2_stats8_kernel<float>): tests + kernel time in the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_train_gpu.py -q -k "pillar or stem or bit_reproducible or gradients_vs_oracle" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_g47 -o st -- python bench.py --lean --steps 10 --warmup 3 > gpurun_out/r05/g47_run.log 2>&1
find /tmp/pf_g47 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g47_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g47_kernel_stats.csv 13 90 | grep -E "total|pfn_bwd_l2_stats"
