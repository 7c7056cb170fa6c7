# r05 lease 2: full GPU suite after the per-call precision plumbing (P3_F32X3), + lean bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05/g02_suite.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r05/g02_suite.txt
tail -25 gpurun_out/r05/g02_suite.txt
python bench.py --lean 2>&1 | tail -1 > gpurun_out/r05/g02_bench_fp32x3.json
tail -c 400 gpurun_out/r05/g02_bench_fp32x3.json
