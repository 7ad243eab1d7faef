#!/bin/bash
# `python bench.py --gpus 2` (two ranks on GPU 0 over gloo — the command of tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu)
# N times in a row WITHOUT a retry; the complete output of any failed run is kept.  Usage: scripts/two_rank_soak.sh [N] [extra bench flags]
N=${1:-20}; shift
O=gpurun_out/prof; mkdir -p $O
export CPX_BENCH_BACKEND=gloo CPX_BENCH_DEVICE=0 GLOO_SOCKET_IFNAME=lo
unset WORLD_SIZE RANK LOCAL_RANK
ok=0; bad=0
{
  echo "# two_rank_soak: $N x python bench.py --gpus 2 --ell 28 --batch 128 --ctxs 2 --steps 2 --warmup 1 --no-extra-shapes --no-traffic $* (commit ${COMMIT:-unknown})"
  for i in $(seq 1 $N); do
    t0=$(date +%s%N)
    python bench.py --gpus 2 --ell 28 --batch 128 --ctxs 2 --steps 2 --warmup 1 --no-extra-shapes --no-traffic "$@" > /tmp/soak_$i.out 2> /tmp/soak_$i.err
    rc=$?
    t1=$(date +%s%N)
    v=$(grep -o '"value": [0-9.]*' /tmp/soak_$i.out | head -1)
    v="$v dominant $(grep -o '"roofline": {"bound": "hbm", "kernel": "[^"]*"' /tmp/soak_$i.out | head -1 | sed 's/.*"kernel": //') $(grep -o '"errors": {[^}]*}' /tmp/soak_$i.out | head -1)"
    printf "run %2d rc=%d %d.%d s %s\n" $i $rc $(( (t1 - t0) / 1000000000 )) $(( (t1 - t0) / 100000000 % 10 )) "$v"
    if [ $rc -ne 0 ]; then bad=$((bad+1)); { echo "==== run $i rc=$rc stdout"; cat /tmp/soak_$i.out; echo "==== stderr"; cat /tmp/soak_$i.err; } > $O/two_rank_soak_failure_$i.log; else ok=$((ok+1)); fi
  done
  echo "# green $ok / $N, failed $bad"
} | tee $O/two_rank_soak.txt
