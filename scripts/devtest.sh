cd $GRAFT_REPO_ROOT
echo "=== device path, all tests"; CPX_DEVICE_TRANSCRIPT_MIN_BATCH=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_whisk.py -m gpu -x -q -k "not variants and not groupings and not large_batch" 2>&1 | tail -30
