#!/bin/bash
# Rehearsal of the 8-rank host budget on ONE GPU: at N = 8 under a 16-core container quota bench.py gives every engine
# context 2 host threads (effective_cores() // (world * contexts)).  The headline is therefore re-run with --host-threads 2 and the whole
# process pinned to 2 cores per context (taskset), back to back with the default line on the same box; then two ranks on GPU 0 over gloo
# with the same budget (4 contexts, 8 cores).  Output: gpurun_out/prof/${RND}_host_budget.txt
RND=${RND:-r05}
R=$PWD
O=$R/gpurun_out/prof
mkdir -p $O
F=$O/${RND}_host_budget.txt
S="--no-cpu-baseline --no-extra-shapes --synth-cache /tmp/synth.pkl --steps 6 --warmup 1"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-58s %9.1f proofs/s  %7.1f ms/step  n_gpus=%d host_threads_per_context=%s all_verified=%s' % (sys.argv[1], d['value'], d['ms_per_step'], d['n_gpus'], d['config'].get('host_threads_per_context'), d['all_verified']))" "$1"; }
{
echo "# host budget rehearsal, $(nproc) logical CPUs visible, cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
python bench.py $S --steps 1 --warmup 0 > /dev/null 2>&1   # synthetic batch cache
python bench.py $S 2>/dev/null | line "default (host threads by quota, no pinning)"
taskset -c 0-3 python bench.py $S --host-threads 2 2>/dev/null | line "--host-threads 2, taskset 4 cores (2 per context)"
taskset -c 0-1 python bench.py $S --host-threads 2 2>/dev/null | line "--host-threads 2, taskset 2 cores (1 per context)"
python bench.py $S 2>/dev/null | line "default again"
CPX_BENCH_BACKEND=gloo CPX_BENCH_DEVICE=0 taskset -c 0-7 python bench.py --gpus 2 --batch 8192 $S --host-threads 2 2>/dev/null | line "2 ranks on GPU 0 (gloo), 8192 per rank, 2 thr/ctx, 8 cores"
CPX_BENCH_BACKEND=gloo CPX_BENCH_DEVICE=0 python bench.py --gpus 2 --batch 8192 $S 2>/dev/null | line "2 ranks on GPU 0 (gloo), 8192 per rank, default threads"
} > $F 2>&1
cat $F
