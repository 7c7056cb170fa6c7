# r05 lease 3: first run of the planes kernels: op tests, microbench, model parity in fp32x3, lean bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_x3_gpu.py -x -q > gpurun_out/r05/g03_x3_tests.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r05/g03_x3_tests.txt
tail -30 gpurun_out/r05/g03_x3_tests.txt
timeout 300 python tools/mb_x3.py > gpurun_out/r05/g03_mb_x3.txt 2>&1
cat gpurun_out/r05/g03_mb_x3.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "fp32x3 or two_models" > gpurun_out/r05/g03_model_tests.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r05/g03_model_tests.txt
tail -8 gpurun_out/r05/g03_model_tests.txt
timeout 600 python bench.py --lean 2>&1 | tail -1 > gpurun_out/r05/g03_bench_fp32x3.json
python -c "import json; d=json.load(open('gpurun_out/r05/g03_bench_fp32x3.json')); print('fp32x3 ms/step', d['ms_per_step'], d['value'])"
