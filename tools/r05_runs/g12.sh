cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_train_gpu.py -q -k "bit_reproducible or raises" > gpurun_out/r05/g12_det.txt 2>&1
grep -E "passed|failed|AssertionError|^E  " gpurun_out/r05/g12_det.txt | cut -c1-1500 | head -20
timeout 900 python -m pytest tests/test_backward_gpu.py tests/test_model_gpu.py -x -q -k "pillar or train_step or fusion-fp32" 2>&1 | tail -3
