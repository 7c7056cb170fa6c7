# r05: fused PFN layer with the next pillar prefetched: tests, dense leg, 40 k + 3 k profiles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_pillar_membership_gpu.py tests/test_backward_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -x -q -k "pillar or canvas or lidar or stem or bit_reproducible" > gpurun_out/r05/g38_tests.txt 2>&1
tail -3 gpurun_out/r05/g38_tests.txt | cut -c1-250
python bench.py --no-cpu-baseline --no-fp32-leg --no-predict --no-host-feed --no-ffl --no-kernel-timing 2>&1 | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step']); print(json.dumps(d.get('dense_lidar'))[200:900])"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_d40 -o st -- python bench.py --lean --points 40000 --steps 6 --warmup 2 > gpurun_out/r05/g38_run.log 2>&1
find /tmp/pf_d40 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g38_dense40k_fp32x3_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g38_dense40k_fp32x3_kernel_stats.csv 9 70 > gpurun_out/r05/g38_dense40k_fp32x3_summary.txt
grep -E "total|pfn|pillar" gpurun_out/r05/g38_dense40k_fp32x3_summary.txt
grep '"metric"' gpurun_out/r05/g38_run.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 40k ms/step', d['ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_3k -o st -- python bench.py --lean --steps 10 --warmup 3 > gpurun_out/r05/g38_run3k.log 2>&1
find /tmp/pf_3k -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g38_3k_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g38_3k_kernel_stats.csv 13 90 | grep -E "total|pillar_sort"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_fwd -o st -- python tools/prof_dense_fwd.py 40000 fp32x3 > gpurun_out/r05/g38_fwd.log 2>&1
grep "points per tile" gpurun_out/r05/g38_fwd.log
find /tmp/pf_fwd -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g38_fwd_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g38_fwd_kernel_stats.csv 13 30 | cut -c1-130
