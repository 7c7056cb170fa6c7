# r05: pfn_bwd_l1 with four rows per parity in flight: tests + 40 k-point kernel times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_backward_gpu.py tests/test_train_gpu.py -x -q -k "pillar or bit_reproducible or stem" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_d40 -o st -- python bench.py --lean --points 40000 --steps 6 --warmup 2 > gpurun_out/r05/g28_run.log 2>&1
find /tmp/pf_d40 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g28_dense40k_fp32x3_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g28_dense40k_fp32x3_kernel_stats.csv 9 70 | grep -E "total|pfn|pillar"
