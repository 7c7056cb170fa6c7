# full GPU suite + a lean bench line on one box: bash tools/gpu_suite.sh <tag>     (the suite is bounded: a hang must not eat the box's time limit)
T=${1:-run}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04/suite_$T.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r04/suite_$T.txt
tail -40 gpurun_out/r04/suite_$T.txt
python bench.py --no-cpu-baseline --no-fp32-leg --no-predict --no-host-feed --no-ffl 2>&1 | tail -1 > gpurun_out/r04/bench_$T.json
python -c "import json; d=json.load(open('gpurun_out/r04/bench_$T.json')); print('ms/step', d['ms_per_step'], 'fwd', d['fwd_ms_per_batch'], 'enc', d['encoder_fwd']['ms_per_batch'], d['encoder_fwd']['mfma_frac'], 'roof', d['roofline']['kernel'], d['roofline']['avg_launch_us'])"
