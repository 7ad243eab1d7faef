cd $GRAFT_REPO_ROOT
run() { echo "=== $*"; env "$@" timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-shapes --synth-cache /tmp/synth.pkl $EXTRA 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(round(d['value']), round(d['ms_per_step']), d['all_verified'])
"; }
EXTRA="--ctxs 2" run CPX_PINNED=1
EXTRA="--ctxs 2" run CPX_PINNED=0
EXTRA="--ctxs 2" run CPX_PINNED=1
EXTRA="--ctxs 2" run CPX_PINNED=0
EXTRA="--ctxs 1" run CPX_PINNED=1
EXTRA="--ctxs 1" run CPX_PINNED=0
