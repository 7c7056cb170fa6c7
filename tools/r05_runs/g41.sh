# r05: after the ragged-round rule: model / train / backward / x3 tests, then the round's artefacts again (tools/final_prof_r05.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1800 python -m pytest tests/test_x3_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py tests/test_backward_gpu.py -q > gpurun_out/r05/g41_tests.txt 2>&1
tail -3 gpurun_out/r05/g41_tests.txt | cut -c1-200
bash tools/final_prof_r05.sh > gpurun_out/final_prof_r05.log 2>&1
tail -22 gpurun_out/final_prof_r05.log | cut -c1-300
