#!/bin/bash
# Regenerates the measurement artefacts of profiles/ on a GPU box (run from the repository root).
# rocprofv3 wants a writable cwd/TMPDIR: the profiler runs from /tmp.  PMC passes are separate runs (no traces with --pmc).
set -x
R=$PWD
mkdir -p gpurun_out/prof
python bench.py --synth-cache /tmp/synth.pkl > gpurun_out/prof/bench_default.log 2>&1
tail -1 gpurun_out/prof/bench_default.log > gpurun_out/prof/r01_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --synth-cache /tmp/synth.pkl > $R/gpurun_out/prof/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/prof/r01_kernel_stats.csv
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/scripts/gpu_busy.py $f after_last=k_fix_build from_first=k_compress > $R/gpurun_out/prof/r01_gpu_busy.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc1 -- python $R/bench.py --no-cpu-baseline --synth-cache /tmp/synth.pkl --steps 1 --warmup 0 > $R/gpurun_out/prof/pmc1.log 2>&1
f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1); python $R/scripts/pmc_summary.py $f $R/gpurun_out/prof/r01_pmc_sq.md > /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc2 -- python $R/bench.py --no-cpu-baseline --synth-cache /tmp/synth.pkl --steps 1 --warmup 0 > $R/gpurun_out/prof/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc3 -- python $R/bench.py --no-cpu-baseline --synth-cache /tmp/synth.pkl --steps 1 --warmup 0 > $R/gpurun_out/prof/pmc3.log 2>&1
python $R/scripts/hbm_counters.py $R/gpurun_out/prof/r01_hbm_counters.json 2048 2 $(find /tmp/pmc2 -name "*counter_collection.csv" | head -1) $(find /tmp/pmc3 -name "*counter_collection.csv" | head -1) > $R/gpurun_out/prof/r01_hbm_counters.txt 2>&1
cd $R; cat gpurun_out/prof/r01_bench.json | cut -c1-400; cat gpurun_out/prof/r01_gpu_busy.txt | head -5; cat gpurun_out/prof/r01_hbm_counters.txt
