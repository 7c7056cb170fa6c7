cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py -q -k "parked or deferred or bit_reproducible or poison" 2>&1 | tail -4
python bench.py --lean 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 3k ms/step', d['ms_per_step'])"
