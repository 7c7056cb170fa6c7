# r05: ragged last round of the 128 x 384 planes tile -> whole rounds on the big tile, remainder on the 128 x 128 tile: tests + same-box A/B of the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_x3_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do
  P3_X3_RAGGED=0 python bench.py --lean 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unsplit ms/step', d['ms_per_step'])"
  python bench.py --lean 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split   ms/step', d['ms_per_step'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_g40 -o st -- python bench.py --lean --steps 10 --warmup 3 > gpurun_out/r05/g40_run.log 2>&1
find /tmp/pf_g40 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g40_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g40_kernel_stats.csv 13 12 | cut -c1-130
