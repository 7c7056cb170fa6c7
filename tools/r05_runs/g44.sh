cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hisup_gpu.py tests/test_input_pipeline_gpu.py -q 2>&1 | tail -3
