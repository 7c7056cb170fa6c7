# dense LiDAR (40 k points per tile): kernel stats of the train step in both precisions
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_d40 -o st -- python bench.py --lean --points 40000 --steps 6 --warmup 2 > gpurun_out/r05/g14_run.log 2>&1
find /tmp/pf_d40 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g14_dense40k_fp32x3_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g14_dense40k_fp32x3_kernel_stats.csv 9 70 | grep -E "total|pfn|pillar|gemm_kernel<float, float, 0|gemm_tn_kernel<float, 0|bn_fin|det_reduce" 
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_d40b -o st -- python bench.py --lean --precision bf16 --points 40000 --steps 10 --warmup 2 > gpurun_out/r05/g14_run_bf16.log 2>&1
find /tmp/pf_d40b -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g14_dense40k_bf16_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g14_dense40k_bf16_kernel_stats.csv 13 80 | grep -E "total|pfn|pillar|bn_fin"
tail -1 gpurun_out/r05/g14_run.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 40k ms/step', d['ms_per_step'])"
tail -1 gpurun_out/r05/g14_run_bf16.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 40k ms/step', d['ms_per_step'])"
