cd $GRAFT_REPO_ROOT
python tools/mb_proj.py 2>&1 | tail -9
