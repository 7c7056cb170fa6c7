cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1800 python -m pytest tests/test_train_gpu.py tests/test_backward_gpu.py tests/test_x3_gpu.py tests/test_syncbn_gpu.py -x -q > gpurun_out/r05/g16_tests.txt 2>&1
tail -4 gpurun_out/r05/g16_tests.txt
python bench.py --lean 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 3k ms/step', d['ms_per_step'])"
P3_DEFER_REDUCE=0 python bench.py --lean 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 3k ms/step, immediate reduces', d['ms_per_step'])"
python bench.py --lean 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32x3 3k ms/step', d['ms_per_step'])"
