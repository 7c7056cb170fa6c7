cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_input_pipeline_gpu.py tests/test_ffl_loss_gpu.py tests/test_afm_gpu.py tests/test_train_gpu.py tests/test_syncbn_gpu.py -x -q -m gpu 2>&1 | tail -60
echo "exit: $?"
