cd $GRAFT_REPO_ROOT
python scripts/devtest.py 252 1024 2>&1 | tail -6
python scripts/devtest.py 252 4096 2>&1 | tail -6
