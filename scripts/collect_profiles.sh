#!/bin/bash
# Regenerates the measurement artefacts of profiles/ on a GPU box AT THE COMMIT IT RUNS ON (run from the repository root; results land in
# gpurun_out/prof/, copy them to profiles/).  rocprofv3 wants a writable cwd/TMPDIR: the profiler runs from /tmp.  PMC passes are separate
# runs (counters only, no trace domain beside them).  roofline.traffic is measured by bench.py itself (two --pmc child passes of the run).
#   RND=r06 COMMIT=$(git rev-parse --short HEAD) bash scripts/collect_profiles.sh [quick]
set -x
RND=${RND:-r06}
R=$PWD
O=$R/gpurun_out/prof
S="--synth-cache /tmp/synth.pkl"
LEAN="--no-cpu-baseline --no-extra-shapes --no-traffic"
mkdir -p $O
echo "commit ${COMMIT:-unknown}  $(date -u +%FT%TZ)" > $O/${RND}_collection.txt
python bench.py $LEAN $S --steps 1 --warmup 0 > /dev/null 2>&1   # builds the synthetic batch cache
# 1. the bench line as the driver runs it (20 steps, 5 warm-up): THE line DESIGN.md / README.md quote
python bench.py $S --steps 20 --warmup 5 > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/${RND}_bench.json
cd /tmp && export TMPDIR=/tmp
# 2. kernel statistics of the same command (rocprofv3 --kernel-trace --stats) + the line the profiled process printed
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py $LEAN $S > $O/kt.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/${RND}_kernel_stats.csv
grep -o '{"metric.*' $O/kt.log | head -1 > $O/${RND}_bench_under_rocprof.json
python $R/scripts/gpu_busy.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) after_last=k_fix_build from_first=k_compress > $O/${RND}_gpu_busy.txt 2>&1
python $R/scripts/kernel_stats_md.py $O/${RND}_kernel_stats.csv > $O/${RND}_kernel_stats.md 2>/dev/null
# 3. issue counters (SQ) of one pass of both contexts
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc1 -- python $R/bench.py $LEAN $S --steps 1 --warmup 0 > $O/pmc1.log 2>&1
python $R/scripts/pmc_summary.py $(find /tmp/pmc1 -name "*counter_collection.csv" | head -1) $O/${RND}_pmc_sq.md > /dev/null
cd $R
# 4. in-process interleaved A/Bs of this round's decisions (bench.py --ab: same contexts, same box, blocks of passes alternating)
python bench.py $S --ab table_chunks=2,1 --ab-rounds 6 2>/dev/null | tail -1 > $O/${RND}_ab_table_chunks.json
python bench.py $S --ab late_rounds=0,4 --ab-rounds 4 2>/dev/null | tail -1 > $O/${RND}_ab_late_rounds.json
python bench.py $S --ab transcript_lane_min_batch=1000000000,8192 --ab-rounds 4 2>/dev/null | tail -1 > $O/${RND}_ab_lane_transcript.json
python bench.py $S --ab rs_pairs=0,1 --ab-rounds 4 2>/dev/null | tail -1 > $O/${RND}_ab_rs_pairs.json
python bench.py --batch 1024 --ctxs 1 --ab late_min_batch=2048,1024 --ab-rounds 4 2>/dev/null | tail -1 > $O/${RND}_ab_late_rounds_1024_proofs.json
python bench.py --batch 128 --ctxs 1 --ab fused_rounds_max=0,256 --ab-rounds 6 2>/dev/null | tail -1 > $O/${RND}_ab_fused_rounds_128_proofs.json
[ "$1" = quick ] && exit 0
# 5. config 5 across the contexts of one GPU, one context alone, every kernel alone (serial_streams)
python bench.py --fused-verify $S 2>/dev/null | tail -1 > $O/${RND}_bench_config5.json
python bench.py $S $LEAN --ctxs 1 --batch 8192 --steps 4 2>/dev/null | tail -1 > $O/${RND}_bench_one_context.json
python bench.py --synth-cache /tmp/synth2.pkl $LEAN --ctxs 2 --batch 16384 --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/${RND}_bench_two_contexts.json   # round 5's default
python bench.py $S $LEAN --ctxs 1 --batch 8192 --steps 4 --opt serial_streams=1 2>/dev/null | tail -1 > $O/${RND}_bench_serial_streams.json
# 6. lone proof: latency, and the kernel timeline launch by launch
python scripts/single_proof_latency.py > $O/${RND}_single_proof_latency.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/scripts/single_proof_timeline.py run > /dev/null 2>&1; python $R/scripts/single_proof_timeline.py report /tmp/tl > $O/${RND}_single_proof_timeline.txt 2>&1 )
# 7. literal small shapes
{ for b in 32 64 128 256 512 1024; do python scripts/shape_sweep.py $b 2>&1 | tail -1; done; } > $O/${RND}_shape_128.txt
# 8. the two-rank launch, twenty times, no retry
bash scripts/two_rank_soak.sh 20 > /dev/null 2>&1
cp $O/two_rank_soak.txt $O/${RND}_two_rank_soak.txt
cut -c1-300 $O/${RND}_bench.json; head -3 $O/${RND}_gpu_busy.txt
