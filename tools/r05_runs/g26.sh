# r05: the final commit on a fresh box - smoke(), the ScoreNet kernel tests, one lean bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_ops_gpu.py -q -k "x3 or scorenet or pair or dual or rows" 2>&1 | tail -2
python bench.py --lean 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 3k ms/step', d['ms_per_step'], 'tiles/s', d['value'])"
