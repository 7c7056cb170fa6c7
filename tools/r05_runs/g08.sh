# r05 lease 8: rocprofv3 kernel stats of the fp32x3 step (planes stack), stack on / off A-B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_x3 -o stx3 -- python bench.py --lean --steps 6 --warmup 2 > gpurun_out/r05/g08_run.log 2>&1
find /tmp/pf_x3 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g08_fp32x3_step_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r05/g08_fp32x3_step_kernel_stats.csv 9 45 > gpurun_out/r05/g08_fp32x3_step_summary.txt
cat gpurun_out/r05/g08_fp32x3_step_summary.txt
P3_X3_STACK=0 python bench.py --lean 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stack off: ms/step', d['ms_per_step'])"
python bench.py --lean 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stack on : ms/step', d['ms_per_step'])"
