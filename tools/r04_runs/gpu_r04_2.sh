set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_pillar_membership_gpu.py -x -q -m gpu 2>&1 | tail -60 > gpurun_out/r04/t1a.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -60 > gpurun_out/r04/t1b.log
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "sinkhorn" 2>&1 | tail -30 > gpurun_out/r04/t5.log
python tools/mb_sinkhorn.py > gpurun_out/r04/mb_sinkhorn.txt 2>&1
for b in 64 32 16 8; do
  python bench.py --steps 10 --batch $b --no-cpu-baseline --no-fp32-leg --no-predict --no-host-feed --no-ffl --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['encoder_fwd']; print('batch', d['config']['tiles_per_gpu'], 'step ms', d['ms_per_step'], 'tiles/s', d['value'], 'fwd ms', d['fwd_ms_per_batch'], 'enc ms', e['ms_per_batch'], 'enc tiles/s', e['tiles_per_s'], 'enc frac', e['mfma_frac'])" >> gpurun_out/r04/batch_sweep.txt 2>&1
done
cat gpurun_out/r04/t1a.log gpurun_out/r04/t1b.log | tail -80
cat gpurun_out/r04/t5.log gpurun_out/r04/mb_sinkhorn.txt gpurun_out/r04/batch_sweep.txt
