cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_pillar_membership_gpu.py tests/test_backward_gpu.py tests/test_model_gpu.py -q -k "pillar or canvas or lidar or stem" 2>&1 | tail -2
