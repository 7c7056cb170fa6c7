# r05 lease 4: planes kernels: op tests, SQ counters of the microbenchmark (what bounds them), model tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_x3_gpu.py -q > gpurun_out/r05/g04_x3_tests.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r05/g04_x3_tests.txt
grep -E "passed|failed|FAILED|Error|assert " gpurun_out/r05/g04_x3_tests.txt | head -30
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pf_sq -o q -- python tools/mb_x3.py > gpurun_out/r05/g04_sq_run.log 2>&1
python tools/pmc_kernels.py /tmp/pf_sq "x3" > gpurun_out/r05/g04_sq_x3.txt 2>&1
cat gpurun_out/r05/g04_sq_x3.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/pf_sq2 -o q -- python tools/mb_x3.py > gpurun_out/r05/g04_sq2_run.log 2>&1
python tools/pmc_kernels.py /tmp/pf_sq2 "x3" > gpurun_out/r05/g04_sq2_x3.txt 2>&1
cat gpurun_out/r05/g04_sq2_x3.txt | head -60
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "two_models" 2>&1 | tail -3
