#!/bin/bash
# Regenerates the measurement artefacts of profiles/ on a GPU box (run from the repository root; results land in
# gpurun_out/prof/, copy them to profiles/).  rocprofv3 wants a writable cwd/TMPDIR: the profiler runs from /tmp.
# PMC passes are separate runs (no traces together with --pmc).  The HBM counter pass runs first because bench.py reads
# profiles/${RND}_hbm_counters.json for roofline.traffic.
set -x
RND=${RND:-r05}
R=$PWD
O=$R/gpurun_out/prof
mkdir -p $O
python bench.py --no-cpu-baseline --no-extra-shapes --synth-cache /tmp/synth.pkl --steps 1 --warmup 0 > /dev/null 2>&1   # builds the synthetic batch cache
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc2 -- python $R/bench.py --no-cpu-baseline --no-extra-shapes --synth-cache /tmp/synth.pkl --steps 1 --warmup 0 > $O/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc3 -- python $R/bench.py --no-cpu-baseline --no-extra-shapes --synth-cache /tmp/synth.pkl --steps 1 --warmup 0 > $O/pmc3.log 2>&1
python $R/scripts/hbm_counters.py $O/${RND}_hbm_counters.json 16384 2 $(find /tmp/pmc2 -name "*counter_collection.csv" | head -1) $(find /tmp/pmc3 -name "*counter_collection.csv" | head -1) ${COMMIT:-unknown} > $O/${RND}_hbm_counters.txt 2>&1
cp $O/${RND}_hbm_counters.json $R/profiles/${RND}_hbm_counters.json
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc1 -- python $R/bench.py --no-cpu-baseline --no-extra-shapes --synth-cache /tmp/synth.pkl --steps 1 --warmup 0 > $O/pmc1.log 2>&1
python $R/scripts/pmc_summary.py $(find /tmp/pmc1 -name "*counter_collection.csv" | head -1) $O/${RND}_pmc_sq.md > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-extra-shapes --synth-cache /tmp/synth.pkl > $O/kt.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/${RND}_kernel_stats.csv
grep -o '{"metric.*' $O/kt.log | head -1 > $O/${RND}_bench_under_rocprof.json   # the bench line of the SAME process the kernel statistics come from
python $R/scripts/gpu_busy.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) after_last=k_fix_build from_first=k_compress > $O/${RND}_gpu_busy.txt 2>&1
cd $R
python $R/scripts/kernel_stats_md.py $O/${RND}_kernel_stats.csv > $O/${RND}_kernel_stats.md 2>/dev/null
python bench.py --synth-cache /tmp/synth.pkl --steps 10 --warmup 2 > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/${RND}_bench.json
python bench.py --fused-verify --synth-cache /tmp/synth.pkl 2>/dev/null | tail -1 > $O/${RND}_bench_config5.json
# one engine context alone (4096 and 8192 proofs per pass): uncontended kernel durations, what a pass consists of
python bench.py --synth-cache /tmp/synth.pkl --no-cpu-baseline --no-extra-shapes --ctxs 1 --batch 8192 --steps 4 2>/dev/null | tail -1 > $O/${RND}_bench_one_context.json
python bench.py --synth-cache /tmp/synth.pkl --no-cpu-baseline --no-extra-shapes --opt late_rounds=0 --steps 6 2>/dev/null | tail -1 > $O/${RND}_bench_late_rounds_off.json
# one context with its side stream's kernels in line on the main stream: every kernel alone on the GPU, a pass = the sum of its kernels
python bench.py --synth-cache /tmp/synth.pkl --no-cpu-baseline --no-extra-shapes --ctxs 1 --batch 8192 --steps 4 --opt serial_streams=1 2>/dev/null | tail -1 > $O/${RND}_bench_serial_streams.json
# A/B of this round's defaults on the same box: the 32-lane transcript prefix kernel at every batch size, phase 1 in one piece
python bench.py --synth-cache /tmp/synth.pkl --no-cpu-baseline --no-extra-shapes --opt transcript_lane_min_batch=1000000000 --steps 6 2>/dev/null | tail -1 > $O/${RND}_bench_wave_transcript.json
python bench.py --synth-cache /tmp/synth.pkl --no-cpu-baseline --no-extra-shapes --steps 6 2>/dev/null | tail -1 > $O/${RND}_bench_defaults_again.json
python scripts/single_proof_latency.py > $O/${RND}_single_proof_latency.txt 2>&1
# the lone-proof timeline (which kernel a lone proof waits for, launch by launch) and the micro-benchmarks behind the latency kernels
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/scripts/single_proof_timeline.py run > /dev/null 2>&1; python $R/scripts/single_proof_timeline.py report /tmp/tl > $O/${RND}_single_proof_timeline.txt 2>&1 )
for m in ${MICRO:-}; do   # MICRO="quad_micro lat_micro place_micro keccak_micro" to rerun the micro-benchmarks (unchanged since r02)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I curdleproofs_amd/csrc scripts/micro/$m.hip -o /tmp/$m > /dev/null 2>&1 && timeout 120 /tmp/$m > $O/${RND}_$m.txt 2>&1
done
# literal shapes, and the A/B of this round's small-batch path on the same box: fused log rounds (round.hip), the quad-per-element T_2 scalar
# multiplications and B as one fixed-base MSM against the round-4 chain of separate kernels
{
  for b in 32 64 128 160 256 1024; do python scripts/shape_sweep.py $b 2>&1 | tail -1; done
  echo "# A/B: the chain of separate kernels (round 4's path): fused rounds off, one-lane k_smul"
  for b in 32 64 128 160; do CPX_FUSED_ROUNDS_MAX=0 CPX_FUSED_SMSM_MAX=0 CPX_SMUL_QUAD_MAX=0 python scripts/shape_sweep.py $b 2>&1 | tail -1; done
  echo "# A/B at 128 proofs, one change at a time"
  CPX_FUSED_SMSM_MAX=0 python scripts/shape_sweep.py 128 2>&1 | tail -1
  CPX_FUSED_COMBINE=0 python scripts/shape_sweep.py 128 2>&1 | tail -1
  CPX_SMUL_QUAD_MAX=0 python scripts/shape_sweep.py 128 2>&1 | tail -1
  CPX_TRANSCRIPT_EXCL_MAX=0 python scripts/shape_sweep.py 128 2>&1 | tail -1
  echo "# A/B at 1024 proofs: wave-per-set reductions (the default up to 1024 proofs since this round) against the work-efficient kernels"
  CPX_REDUCE_WAVE_MAX=2048 python scripts/shape_sweep.py 1024 2>&1 | tail -1
} > $O/${RND}_shape_128.txt
# kernel timeline of one 128-proof pass, launch by launch: this round's path, then the round-4 chain on the same box
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl128 -- python $R/scripts/batch_timeline.py run 128 > /dev/null 2>&1
  { echo "#### 128 proofs, default options: one launch per log round (k_round_fused), k_smul_quad, B as a fixed-base MSM"; python $R/scripts/batch_timeline.py report /tmp/tl128; } > $O/${RND}_batch128_timeline.txt 2>&1
  export CPX_FUSED_ROUNDS_MAX=0 CPX_FUSED_SMSM_MAX=0 CPX_SMUL_QUAD_MAX=0
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl128b -- python $R/scripts/batch_timeline.py run 128 > /dev/null 2>&1
  { echo; echo "#### 128 proofs, fused_rounds_max = fused_smsm_max = smul_quad_max = 0: the chain of separate kernels (round 4's path), same box"; python $R/scripts/batch_timeline.py report /tmp/tl128b; } >> $O/${RND}_batch128_timeline.txt 2>&1 )
bash scripts/host_budget.sh > /dev/null 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I curdleproofs_amd/csrc scripts/micro/affine_micro.hip -o /tmp/affine_micro > /dev/null 2>&1 && timeout 300 /tmp/affine_micro 16 > $O/${RND}_affine_micro_raw.txt 2>&1
cut -c1-300 $O/${RND}_bench.json; head -3 $O/${RND}_gpu_busy.txt
