# r05: the ScoreNet's fp32x3 kernels (pair_bwd_x3, pair_fwd_x3, rows_x3, pair_dw_x3, mask2_dw_x3): parity tests, ScoreNet / model tests, kernel times in the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_ops_gpu.py -q -k "x3 or scorenet or pair or dual or rows" > gpurun_out/r05/g18_tests.txt 2>&1
tail -25 gpurun_out/r05/g18_tests.txt | cut -c1-300
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -x -q > gpurun_out/r05/g18_model.txt 2>&1
tail -6 gpurun_out/r05/g18_model.txt | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_g18 -o st -- python bench.py --lean --steps 10 --warmup 3 > gpurun_out/r05/g18_run.log 2>&1
find /tmp/pf_g18 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g18_fp32x3_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g18_fp32x3_kernel_stats.csv 13 60 > gpurun_out/r05/g18_fp32x3_summary.txt
head -30 gpurun_out/r05/g18_fp32x3_summary.txt | cut -c1-160
grep -E "pair|mask2|rows|row_affine|score" gpurun_out/r05/g18_fp32x3_summary.txt | cut -c1-160
tail -1 gpurun_out/r05/g18_run.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 3k ms/step (profiled)', d['ms_per_step'])"
python bench.py --lean 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 3k ms/step', d['ms_per_step'])"
