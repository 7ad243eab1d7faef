cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "not variants and not groupings" 2>&1 | tail -4
run() { echo "=== $*"; env "$@" timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-shapes --synth-cache /tmp/synth.pkl $EXTRA 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(round(d['value']), round(d['ms_per_step']), d['all_verified'])
print({k:round(v,1) for k,v in d['kernel_ms_per_step'].items()})
"; }
EXTRA="--ctxs 2" run A=1
EXTRA="--ctxs 1" run A=1
