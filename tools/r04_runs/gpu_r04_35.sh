cd $GRAFT_REPO_ROOT
python tools/mb_wide384.py 2>&1 | tail -10
