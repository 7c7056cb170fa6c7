# r05 lease 6: planes GEMM v3 (software-pipelined 128 x 384 tile) + attention backward writing planes: op tests, microbench, SQ counters, model tests, lean bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_x3_gpu.py -q > gpurun_out/r05/g06_x3_tests.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r05/g06_x3_tests.txt
grep -E "passed|failed|^FAILED" gpurun_out/r05/g06_x3_tests.txt | head -30
timeout 300 python tools/mb_x3.py > gpurun_out/r05/g06_mb_x3.txt 2>&1
cat gpurun_out/r05/g06_mb_x3.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pf_sq -o q -- python tools/mb_x3.py > gpurun_out/r05/g06_sq_run.log 2>&1
python tools/pmc_kernels.py /tmp/pf_sq "x3" > gpurun_out/r05/g06_sq_x3.txt 2>&1
grep -E "kernel|MFMA_BUSY|WAIT_ANY|WAIT_INST|GRBM|BANK" gpurun_out/r05/g06_sq_x3.txt
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_backward_gpu.py -x -q -k "fp32x3 or two_models" > gpurun_out/r05/g06_model_tests.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r05/g06_model_tests.txt
tail -5 gpurun_out/r05/g06_model_tests.txt
timeout 600 python bench.py --lean 2>&1 | tail -1 > gpurun_out/r05/g06_bench_fp32x3.json
python -c "import json; d=json.load(open('gpurun_out/r05/g06_bench_fp32x3.json')); print('fp32x3 ms/step', d['ms_per_step'], d['value'])"
