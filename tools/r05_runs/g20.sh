# r05: ScoreNet fp32x3 kernels, third pass (DMA parking in pair_dw_x3, pipelined fragments in pair_fwd_x3 / pair_dw_x3): tests, kernel times, SQ counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_ops_gpu.py -q -x -k "x3 or scorenet or pair or dual or rows" > gpurun_out/r05/g20_tests.txt 2>&1
tail -3 gpurun_out/r05/g20_tests.txt | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_g20 -o st -- python bench.py --lean --steps 10 --warmup 3 > gpurun_out/r05/g20_run.log 2>&1
find /tmp/pf_g20 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g20_fp32x3_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g20_fp32x3_kernel_stats.csv 13 70 > gpurun_out/r05/g20_fp32x3_summary.txt
head -3 gpurun_out/r05/g20_fp32x3_summary.txt | cut -c1-160
grep -E "pair|mask2|rows_x3" gpurun_out/r05/g20_fp32x3_summary.txt | cut -c1-160
grep '"metric"' gpurun_out/r05/g20_run.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 3k ms/step (profiled)', d['ms_per_step'])"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pf_sq20 -o q -- python bench.py --lean --graph 0 --steps 3 --warmup 2 > gpurun_out/r05/g20_sq_run.log 2>&1
python tools/pmc_kernels.py /tmp/pf_sq20 > gpurun_out/r05/g20_sq_counters.txt 2>&1
grep -A9 -E "^pair_|^mask2_dw_x3|^rows_x3" gpurun_out/r05/g20_sq_counters.txt | cut -c1-120
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/pf_sq20b -o q -- python bench.py --lean --graph 0 --steps 3 --warmup 2 > gpurun_out/r05/g20_sq_run_b.log 2>&1
python tools/pmc_kernels.py /tmp/pf_sq20b > gpurun_out/r05/g20_sq_counters_b.txt 2>&1
grep -A9 -E "^pair_|^mask2_dw_x3" gpurun_out/r05/g20_sq_counters_b.txt | cut -c1-120
