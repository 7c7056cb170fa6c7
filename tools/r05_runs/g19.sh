# r05: ScoreNet fp32x3 kernels, second pass (pair_bwd_x3 with DMA parking + pipelined halves): tests + kernel times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_ops_gpu.py -q -x -k "x3 or scorenet or pair or dual or rows" > gpurun_out/r05/g19_tests.txt 2>&1
tail -5 gpurun_out/r05/g19_tests.txt | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_g19 -o st -- python bench.py --lean --steps 10 --warmup 3 > gpurun_out/r05/g19_run.log 2>&1
find /tmp/pf_g19 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g19_fp32x3_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g19_fp32x3_kernel_stats.csv 13 70 > gpurun_out/r05/g19_fp32x3_summary.txt
head -3 gpurun_out/r05/g19_fp32x3_summary.txt | cut -c1-160
grep -E "pair|mask2|rows|row_affine|score" gpurun_out/r05/g19_fp32x3_summary.txt | cut -c1-160
tail -1 gpurun_out/r05/g19_run.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 3k ms/step (profiled)', d['ms_per_step'])"
