cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "two_ranks or crs_hex" 2>&1 | tail -15
