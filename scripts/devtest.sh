cd $GRAFT_REPO_ROOT
echo "=== device path trace"; CPX_TRACE=1 CPX_DEVICE_TRANSCRIPT_MIN_BATCH=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "test_prove_verify_ell28_readme_example" 2>&1 | grep -c "cpx trace"
echo "=== device path, more tests"; CPX_DEVICE_TRANSCRIPT_MIN_BATCH=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_whisk.py -m gpu -x -q -k "prove or whisk or verify" 2>&1 | tail -30
