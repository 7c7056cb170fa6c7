cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python tools/diag_determinism.py fp32x3 > gpurun_out/r05/g10_det_fp32x3.txt 2>&1; cat gpurun_out/r05/g10_det_fp32x3.txt | tail -40
python tools/diag_determinism.py fp32 > gpurun_out/r05/g10_det_fp32.txt 2>&1; tail -12 gpurun_out/r05/g10_det_fp32.txt
timeout 900 python -m pytest tests/test_pillar_membership_gpu.py tests/test_backward_gpu.py -q -k "dense_cap or pillar_stem_backward" 2>&1 | tail -5
