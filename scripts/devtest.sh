cd $GRAFT_REPO_ROOT
run() { echo "=== $*"; env "$@" timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --synth-cache /tmp/synth.pkl $EXTRA 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(round(d['value']), round(d['ms_per_step']))
print({k:round(v,1) for k,v in d['kernel_ms_per_step'].items() if k.startswith('k_msm')})
"; }
EXTRA="--ctxs 1" run A=1
EXTRA="--ctxs 1" run CPX_LIB=$PWD/curdleproofs_amd/_lib/libcpx_nopad.so
EXTRA="--ctxs 2" run A=1
EXTRA="--ctxs 2" run CPX_LIB=$PWD/curdleproofs_amd/_lib/libcpx_nopad.so
