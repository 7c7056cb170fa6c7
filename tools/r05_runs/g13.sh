cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_train_gpu.py -q -k "bit_reproducible" 2>&1 | grep -E "passed|failed|AssertionError" | cut -c1-600 | head
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_syncbn_gpu.py -x -q -k "pillar_stem or two_ranks" 2>&1 | tail -3
python bench.py --lean 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 ms/step', d['ms_per_step'])"
