// Task descriptors and launchers of the gfx950 kernels (kernels.hip) — shared with the host engine.
#pragma once
#include <hip/hip_runtime.h>
#include "g1.hpp"
#include "g1_28.hpp"

namespace cpx {

enum : uint32_t {
  MSM_SCALARS_CANONICAL = 1u,
  // SmulTask::flags: multiply by the plain non-adjacent form of the 255-bit scalar (257 doublings) instead of the endomorphism split.  The split
  // k P = t P + q (-phi(P)) is an identity of the order-r subgroup only; points of E(Fp) outside it (the cofactor multiplication of a
  // hash-to-curve CRS, tests/crs.rs:38) need this form.
  SMUL_PLAIN = 2u
};

// Tunables of the launchers and of the host engine.  They belong to an engine context (cpx_ctx_set_option, include/cpx.h): two
// contexts of one process may run different settings.  The environment (CPX_<NAME>) only provides the defaults a new context
// starts from (default_options(), read once per process).  Keys are the lower-case field names.
struct Options {
  // kernel selection (kernels.hip); *_max: largest launch that still takes the latency (wave / quad per item) form of a kernel
  long tail_wave_max = 256;
  long tbw_min_waves = 1024;       // k_msm_tblw: smallest grid that still gets the widest window grouping (one wave per SIMD: 2048 cost 1-5 % between 128 and 384 proofs)
  long tbw_wpw = 0;                // 2|4|8|16|32 pins the windows per wave of k_msm_tblw (0 = by grid size)
  long tbw_lat_waves = 2048;
  long tbw_slices = 0;             // 1|2|4 pins the point slices of the bucket-list MSM (0 = by grid size)
  long reduce_wave_max = -1;       // -1 = by the batch in flight (16384 sets up to 512 proofs, 2048 above)
  long fix_min_waves = 4096;
  long fix_lat_waves = 2048;
  long finalize_wave_max = 512;
  long table_quad_max = 4096;
  long decompress_quad_max = 2048;
  // host engine (engine.cpp)
  long fix_bits = 16;              // radix of the fixed-base CRS table of multiples: 16 (17.5 GB at ell = 252), 8 (0.1 GB) or 19 (14 windows, 122 GB); read by set_crs
  long device_min_batch = 56;      // batches of at least this many proofs run the whole protocol on the device (ell = 252: host-driven 22.1 / 25.5 ms per pass at 48 / 64 proofs, device-resident 23.0 / 24.4)
  long spin_team = 1;              // 0: no spinning helper threads for small batches
  long spin_team_threads = 8;
  long inline_below = 16;          // host loops shorter than this run inline
  long cu_mask = 1;                // 0: plain instead of CU-masked streams for the lone-proof prover
  long trace = 0;                  // 1: print the challenges of proof 0 to stderr (debugging aid)
  long late_rounds = 4;            // != 0: the last log2(m) log rounds of a large batch run on m materialised folded bases per family (late.hip); 0: all-MSM everywhere
  long late_m = 0;                 // m = 16 | 32 | 64 (four / five / six late rounds); 0 = by the proof size: 16 up to n = 256, 32 from n = 512 on (ell = 1020: 3.83 -> 4.05 k proofs/s, ell = 508: 8.18 -> 8.61 k)
  long late_min_batch = 2048;      // ... for batches of at least this many proofs (n <= 256; scaled by 256 / n above: 512 at ell = 1020); the lane-per-output kernels need a full GPU to pay
  long late_slices = 8;            // lanes per cross term of a late round (1, 2, 4, 8): 15.3 k proofs/s at 8, 14.9 k at 4 and 2, 14.3 k at 1
  long rs_pairs = 1;               // 1: the prover's R and S MSMs of a proof share their waves (same scalars: one digit sort for both); 0: two separate tasks
  long table_chunks = 0;           // launches of the device prover's per-proof table build: 0 = all rows at once while the scratch stays below 16 GiB (8192 proofs of ell = 252: 13.5 GB), k = exactly k equal chunks (2: round 5's default, 6.7 GB, 0.5 % slower in the round-6 A/B)
  long table_stream_max = 4096;            // up to this many proofs (n <= 256; scaled by 256 / n above) the prover's per-proof tables and B_t, B_u run on a table stream beside phase 1 (engine_device.cpp); larger batches fill the GPU with either and keep them in line
  long transcript_excl_max = 256;          // up to this many proofs a wave of k_transcript_step1 claims the whole register file of its SIMD: no wave of a concurrent kernel is placed beside it (a shared SIMD costs a lone dependent chain 1.7x)
  long transcript_lane_min_batch = 8192;   // batches of at least this many proofs (n <= 256; scaled by n / 256 above) hash the transcript prefix with one LANE per proof (k_transcript_step1_lane): a fifth of the wave instructions but a chain of 19 ms at ell = 252 — only where the table build (prover) and the decompression (verifier: 18.7 ms per 8192 proofs) hide it; +2.2 % at 2 x 8192 proofs
  long p1_split = 1;               // 1: phase 1 without A (it only needs the prover's randomness) runs before the main stream waits for the transcript prefix, A alone behind it: 128 proofs 41.7 -> 40.6 ms per pass, nothing at 8192
  long serial_streams = 0;         // 1 (measurement aid): the side stream's kernels run in line on the main stream — every kernel's duration is its own
  long bench_field = 12;           // cpx_bench_fpmul: 12 = 32-bit-limb standard form, 28 = 28-bit-limb table form
  long fused_rounds_max = 256;     // batches of up to this many proofs (n <= 256; scaled by 256 / n above) run every log round as ONE launch (round.hip); 0: the chain of
                                   // separate kernels.  The in-wave reductions of a fused round pay while its waves have SIMDs of their own (128 proofs x 8 waves = the
                                   // GPU's 1024 SIMDs; 256 proofs x 4): ms per pass with everything of this round, 64 proofs 24.4 -> 22.1, 128: 29.0 -> 26.7, 192: 34.9 -> 33.7, 256: 38.0 -> 36.0; 512 and above: slower (waves share SIMDs)
  long fused_smsm_max = 256;       // ... and the SameMSM rounds up to this many (their in-wave bucket reductions — 2 x 19 quad rounds per wave — do not pay once waves share SIMDs)
  long fused_fix_wpw = 0;          // windows per fixed-base wave of a fused round: 4 | 8 | 16, 0 = by the batch (4 while 16 waves per proof fit the SIMDs, else 8)
  long fused_tbl_wpw = 16;         // windows per table wave of a fused SameMSM round (8 | 16 | 32)
  long fused_combine = -1;         // 1: a SameMSM wave of a fused round runs a fixed-base block (4 windows) and a table block (16 windows): 8 waves per proof instead of
                                   // 12; 0: separate waves; -1 = by the batch (combined from 86 proofs on: 12 waves per proof no longer find a SIMD each)
  long smul_quad_max = 1024;       // the prover's T_2 scalar multiplications of a batch of up to this many elements (4 per proof) run with a quad per element (k_smul_quad)
  long strict_infinity = 0;        // point encodings with the infinity flag set: 0 = the identity whatever the other 382 bits say (ark-bls12-381 ^0.4 `read_g1_compressed`, as recalled: include/cpx.h), 1 = only 0xc0 || 0^47 (the zcash specification's wording; ark-bls12-381 0.5)
  long msm_endo_min = 1;           // cpx_g1_msm / cpx_g1_msm_jac: from this many points on the endomorphism bucket-list path (k_to_table_endo + k_msm_tblw<2, true>), below it the plain windowed accumulation (k_msm_accw: three dependent launches with doubling chains, 1.35 - 1.5 ms per call at 1 - 63 points against 0.72 - 0.77 on the bucket-list path; the default was 64 until round 6)
  long scale_any_point = 0;        // 1: cpx_g1_scale / cpx_g1_fold take ANY point of E(Fp) (plain 257-step double-and-add, SMUL_PLAIN); 0: points of the order-r subgroup (endomorphism split, 129 steps)
};
const Options& default_options();                               // built-in defaults overridden by CPX_<NAME> environment variables
bool set_option(Options& o, const char* key, long value);       // false: unknown key or value out of range
bool get_option(const Options& o, const char* key, long* value);

// Profiling hook: the next launches of the calling thread bind these events to the kernel's begin / end
// (nullptr, nullptr = plain launches).
void set_launch_events(hipEvent_t start, hipEvent_t stop);
int launches_since_set();
void take_launch_events(hipEvent_t* start, hipEvent_t* stop);   // for launchers outside kernels.hip (late.hip): the pending events, counted as a launch

// One multi-scalar multiplication  sum_i scalars[i] * bases[idx ? idx[i] : i],  i < n.
// All pointers are device pointers.  Scalars are Fr in Montgomery form unless flagged canonical.
struct MsmTask {
  const Aff* bases;
  const uint32_t* idx;   // optional gather list into `bases`
  const Fr* scalars;
  uint32_t n;
  uint32_t flags;
  uint32_t conv_off;     // where this task's bases go in the phase's table-form scratch (n entries)
};

// cnt elements:  out[e] = affine( (add ? add[e] : O) + scalars[e * scalar_stride] * mul[e] )
struct SmulTask {
  const Aff* add;        // nullable
  const Aff* mul;
  Aff* out;
  const Fr* scalars;
  uint32_t scalar_stride;   // 0 = one shared scalar (fold), 1 = per element (rescale)
  uint32_t flags;
};

// ---- table-backed MSM (no doubling tails) ----
// Tables and partial sums are in the table kernels' representation (g1_28.hpp: TAff / TJac, 28-bit limbs).
// A table holds, for every base point P_i, 32 affine copies, copy-major (entry (c, i) at base + c*copy_stride + i):
// 2^(8c) * P_i for c = 0..15 and z^2 * 2^(8(c-16)) * P_i = -phi(2^(8(c-16)) P_i) for c = 16..31.  A scalar is split as
// k = +-(+-|t| + q z^2) (glv.hpp); every radix-256 window of |t| and of q has its own copy, so all window sums carry
// weight 1 and no doubling is left in the MSM.
struct TblSeg {
  const TAff* base;         // copy 0
  const uint32_t* idx;      // optional gather list (indices within a copy)
  uint32_t copy_stride;     // in TAff units
  uint32_t n;
};
struct TblTask {
  TblSeg seg[2];            // bases = seg[0] followed by seg[1]
  const Fr* scalars;        // seg[0].n + seg[1].n scalars
  uint32_t flags;
  uint32_t pad;             // first partial slot of the task; it writes 64 / wpw partials
  const uint32_t* digits;   // optional: 9 words per scalar (32 biased digit bytes + the two half signs), written by k_to_table_endo
};
struct TblTmp {   // table-build scratch: a Jacobian copy and the running product of the previous copies' Z
  TJac p;
  TF pre;
};
// single-wave work-groups owning wpw (2 .. 32) windows each: 32 / wpw waves per task, two raw sets per wave
// -> partial sums d_part[task.pad + j], j < 64 / wpw
int msm_tblw_windows_per_wave(const Options& o, int ntasks);
int msm_tblw_parts(int wpw);   // waves (= partial sums) per task
// slices > 1 (a few tasks in flight): that many waves share a task's points; a task then leaves slices * msm_tblw_parts(wpw) partial sums
int msm_tblw_slices(const Options& o, int ntasks, int wpw, int max_n);   // max_n: points of the largest task
void launch_msm_tblw(const TblTask* d_tasks, int ntasks, int wpw, uint32_t* d_raw, uint32_t* d_raw_slot, hipStream_t s, int slices = 1);
// The MSM waves of k_msm_tblw / k_msm_fix leave their 64 lane accumulators as "raw sets" (raw_set_words() 32-bit words
// each; tblw: 2 sets per wave, fix: 1) plus the partial-sum slot every set belongs to; launch_reduce_sets turns them
// into d_part[slot] (bucket sets of k_msm_tblw: weighted sums; k_msm_fix: plain sums).
size_t raw_set_words();
size_t reduce_mid_per_set();   // TJac entries of launch_reduce_sets' d_mid scratch per set (it holds 16 extended-Jacobian group sums)
// Bucket MSM over caller-supplied bases with the endomorphism split and radix-256 windows (no shifted copies): 16 waves
// per task, 32 raw sets per task; after launch_reduce_sets(weighted) the partial sums part[task*32 + 2w + h] carry the
// weight 2^(8w): launch_msm_tail(part, ..., group 16, shift 8, dup 2) finishes.  d_conv: 2 x (sum of n) TAff scratch
// (task.conv_off in units of points, as for launch_msm_accum), d_digits: 9 words per point (the split scalars' digits,
// recoded once per task instead of once per wave), d_ttasks: ntasks TblTask scratch.
// slices > 1 (msm_tblw_slices): that many waves per window share the task's points; 32 * slices raw sets and partial sums per task,
// part[(task*16 + w) * 2 * slices + d] carry the weight 2^(8w) (tail: dup = 2 * slices).
void launch_msm_endo(const MsmTask* d_tasks, int ntasks, int max_n, TAff* d_conv, uint32_t* d_digits, TblTask* d_ttasks, uint32_t* d_raw, uint32_t* d_raw_slot,
                     hipStream_t s, int slices = 1);
// The same for PAIRS of tasks with identical scalar vectors (d_tasks[2 p] and d_tasks[2 p + 1]: same n, same scalars pointer
// contents — the prover's R = <a, vec_R>, S = <a, vec_S>): one wave per (pair, window) sorts the digits once and walks the bucket
// lists once per task.  Raw sets and partial-sum slots as for 2 * npairs tasks of launch_msm_endo (slices = 1).
void launch_msm_endo_pairs(const MsmTask* d_tasks, int npairs, int max_n, TAff* d_conv, uint32_t* d_digits, TblTask* d_ttasks, uint32_t* d_raw, uint32_t* d_raw_slot,
                           hipStream_t s);
// The first nplain sets are plain sums, the following nweighted ones bucket sets (lower / upper half alternating).
// nproofs: how many proofs the batch in flight holds (0: unknown) — picks between the wave-per-set and the work-efficient kernels
void launch_reduce_sets(const Options& o, const uint32_t* d_raw, const uint32_t* d_raw_slot, int nplain, int nweighted, TJac* d_mid, TJac* d_part, hipStream_t s,
                        int nproofs = 0);   // d_mid: reduce_mid_per_set() TJac per set
// For `rows` rows of `npts` standard-form points (row r at d_src + r*src_row_stride) writes copies 0..C-1 of the
// table (row r at d_table + r*row_stride).  d_tmp: rows*npts*(C-1) TblTmp scratch.
// endo = false: copy c = 2^(256 c / C) P.   endo = true (the tables k_msm_tblw reads): copies 0..C/2-1 = 2^(8c) P and
// copies C/2.. = their endomorphism images -phi(2^(8c) P) = z^2 2^(8c) P (glv.hpp) — half the doublings.
void launch_table_build(const Options& o, const Aff* d_src, size_t src_row_stride, TAff* d_table, int rows, size_t row_stride, int npts, int copy_stride, int copies, bool endo,
                        TblTmp* d_tmp, hipStream_t s, int step_bits = 0);   // step_bits > 0: copy c = 2^(step_bits c) P whatever the copy count

// ---- fixed-base MSM over a table of multiples (CRS bases) ----
// fix_tab[(w*M + (m-1))*nc + i] = m * 2^(c*w) * P_i  (affine), w < W = 256/c windows, m = 1..M = 2^(c-1), i < nc.
// A scalar is W signed radix-2^c digits; digit d of window w selects entry |d| directly: one mixed addition per
// (point, window), no buckets.  One wave handles wpw (8 or 16) windows of a task (lane = window x point slice):
// every lane performs the same number of additions.
struct FixTask {
  const uint32_t* idx;      // optional gather list of base indices (< nc)
  const Fr* scalars;
  uint32_t off;             // first base index when idx == nullptr
  uint32_t n;
  uint32_t flags;
  uint32_t out_first;       // first partial slot of this task; it writes W / wpw partials
};
int msm_fix_windows_per_wave(const Options& o, int ntasks, int cbits);
int msm_fix_windows(int cbits);          // ceil(256 / cbits): 32, 16 or (cbits = 19) 14
int msm_fix_parts(int cbits, int wpw);   // waves (= plain raw sets = partial sums) per task
void launch_msm_fix(const FixTask* d_tasks, int ntasks, const TFix* d_fix_tab, int cbits, int wpw, int nc, uint32_t* d_raw, uint32_t* d_raw_slot, hipStream_t s);
// k_msm_fix<16, 2> and k_msm_tblw<2, false> of one phase in ONE launch (a lone proof: neither fills the GPU)
void launch_msm_fix_tblw(const FixTask* d_ftasks, int nftasks, const TFix* d_fix_tab, int nc, uint32_t* d_fraw, uint32_t* d_fraw_slot, const TblTask* d_ttasks,
                         int nttasks, int slices, uint32_t* d_traw, uint32_t* d_traw_slot, hipStream_t s);
// shifted copies (from launch_table_build with 256/c copies) -> table of multiples; d_tmp: threads * chunk TblTmp
void launch_fix_build(const TAff* d_shift, int nc, int cbits, TFix* d_fix_tab, TblTmp* d_tmp, int chunk, hipStream_t s, int segs = 1);   // d_tmp: segs x threads x chunk
// output o = sum of partials [first[o], first[o] + count[o]) -> standard form, normalised + compressed (+ optional affine scatter)
// d_addends (optional): [n][3] indices into d_aff of affine points (results of earlier phases) to add to output o; ~0u = none
// d_comp_index (optional): output o's compressed bytes go to d_comp + 48 * d_comp_index[o] instead of d_comp + 48 * o
void launch_finalize_ranges(const Options& o, const TJac* d_part, const uint32_t* d_first, const uint32_t* d_count, int n, Aff* d_aff, const uint32_t* d_dst_index,
                            uint8_t* d_comp, hipStream_t s, const uint32_t* d_addends = nullptr, const uint32_t* d_comp_index = nullptr);

// ntasks MSMs: bucket accumulation -> d_wsum[ntasks*64] window sums (Jacobian, table representation) ...
// d_conv: scratch for the table-form copies of all bases (sum of n over the tasks, addressed by task.conv_off);
// max_n: largest task.n of the launch
void launch_msm_accum(const MsmTask* d_tasks, int ntasks, int max_n, TAff* d_conv, TJac* d_wsum, hipStream_t s);
// ... then Horner tails: out[t] = sum_{j<group} 2^(shift*j) in[t*group+j]; (8,4) then (8,32) gives the MSM results.
// Exactly one of d_out (table representation) / d_out_std (standard form, last stage) is non-null.
// d_extra (optional): extra_per_out further partial sums added to output t before it is written
// dup: that many consecutive inputs share a weight (out[t] = sum_j 2^(shift*j) (in[(t*group+j)*dup] + ...)).
void launch_msm_tail(const Options& o, const TJac* d_in, TJac* d_out, Jac* d_out_std, int nout, int group, int shift, hipStream_t s, const TJac* d_extra = nullptr,
                     int extra_per_out = 0, int dup = 1);
// Jacobian -> affine (+ optional scatter through dst_index) and optional 48-byte compressed form.
// d_addend_index (optional): entry g != ~0u names a point of d_aff that is added to input g before it is normalised
void launch_finalize(const Jac* d_in, int n, Aff* d_aff, const uint32_t* d_dst_index, uint8_t* d_comp, hipStream_t s, const uint32_t* d_addend_index = nullptr);
// `batch` rows of n points, row r starting at d_in + r*stride; output dense [batch][n][48]
void launch_compress(const Aff* d_in, int n, int stride, int batch, uint8_t* d_out, hipStream_t s, int out_stride = 0);   // out_stride (points per output row), 0 = dense
// `batch` strided rows; only the columns d_cols[0..ncols) are compressed, each to the same column of the output row
void launch_compress_cols(const Aff* d_in, const uint32_t* d_cols, int ncols, int stride, int batch, uint8_t* d_out, int out_stride, hipStream_t s);
// exclusive_simd (the T_2 commitments of a small batch on the side stream, at most 16 waves): every wave claims the whole register file of its SIMD
// quad_max > 0: launches of at most that many elements run with a QUAD per element (k_smul_quad: a third of the one-lane chain's latency, four times its lanes);
// only for tasks without SMUL_PLAIN
void launch_smul(const SmulTask* d_tasks, int ntasks, int cnt, hipStream_t s, bool exclusive_simd = false, long quad_max = 0);
// compressed bytes -> affine; status[i] = 0 ok, 1 malformed / not on curve, 2 not in the r-torsion subgroup
// d_src_offset (optional): encoding i starts at byte d_src_offset[i] of d_in (points inside serialized proofs) instead of 48 i
void launch_decompress(const Options& o, const uint8_t* d_in, int n, Aff* d_out, const uint32_t* d_dst_index, uint8_t* d_status, int check_subgroup, hipStream_t s,
                       const uint32_t* d_src_offset = nullptr);
// Fr algebra of the prover's log rounds on device-resident vectors (kernels.hip: "Fr algebra of the log rounds"):
// IPA vec [nproofs][4][n] = c | d | SG | SGp, out [nproofs][4 (n/2) + 2]; SameMSM vec [nproofs][2][n] = x | SM, out [nproofs][n];
// gam [nproofs][2] = gamma, gamma^-1
// active (0 = n): the leading elements the round still works on — m for the late rounds on materialised bases (late.hip), which
// restart with unit fold coefficients inside the first m entries; the output rows then hold 4 (active/2) + 2 resp. active scalars
void launch_ipa_round_scalars(const Fr* d_vec, int nproofs, int n, int half, const Fr* d_beta, Fr* d_out, hipStream_t s, int active = 0);
void launch_ipa_round_fold(Fr* d_vec, int nproofs, int n, int half, const Fr* d_gam, hipStream_t s, int active = 0);
void launch_smsm_round_scalars(const Fr* d_vec, int nproofs, int n, int half, Fr* d_out, hipStream_t s, int active = 0);
void launch_smsm_round_fold(Fr* d_vec, int nproofs, int n, int half, const Fr* d_gam, hipStream_t s, int active = 0);
// transcript prefix of every proof: absorbs the compressed instance ([nproofs][4*ell*48] + [nproofs][48] for M), draws vec_a;
// d_state: [nproofs][27] u64 (25 STROBE lanes, pos, pos_begin), d_vec_a: [nproofs][ell] Fr
// lane_per_proof: one lane instead of 32 per transcript (large batches: a fifth of the wave instructions, a longer chain)
void launch_transcript_step1(const uint8_t* d_inst_comp, const uint8_t* d_m_comp, int nproofs, int ell, uint64_t* d_state, Fr* d_vec_a, hipStream_t s,
                             bool lane_per_proof = false, bool exclusive_simd = false);
// ---- late log rounds of the batch prover on materialised folded bases (late.hip) ----
// Families of materialised points, per proof [LATE_FAMILIES][m]: G, G' = u o G (IPA), G_b, T_b, U_b (SameMSM)
enum { LATE_F_G = 0, LATE_F_GP, LATE_F_T, LATE_F_U, LATE_F_GB, LATE_FAMILIES };   // layout [family][proof][m]
struct LateOut {
  int fam, hi;              // family; 1: the points whose index has bit `half` set, 0: clear
  uint32_t scal_off;        // the output's m/2 scalars inside the proof's scalar row
};
struct LateRound {
  int m, half, nout, total;           // total = nproofs * nout * slices lanes
  int nproofs, slices;                // slices (1, 2, 4, 8): lanes per output, each taking 32 / slices of the 4-bit windows
  LateOut o[6];
  const Fr* scal;                     // [nproofs][scal_proof_stride]
  uint32_t scal_proof_stride;
  const TAff* tab;                    // [LATE_FAMILIES][nproofs][m][late_tab_entries()]
  TJac* out;                          // [total]: partial sums, slice j of an output carries the weight 2^(128 / slices * j)
};
size_t late_tab_entries();            // table entries per materialised point (8 multiples + their 8 endomorphism images)
size_t late_tmp_per_lane();           // TblTmp scratch entries per point of launch_late_tables (rounded up to 64 points)
bool late_supported(int n, int m);
// out[p * ostride + i] = sum_t scal[p * sstride + t m + i] * CRS column (cols ? cols[t m + i] : t m + i),  i < m
void launch_late_fix(const Fr* d_scal, size_t sstride, const uint32_t* d_cols, int n, int m, const TFix* d_fix_tab, int cbits, int nc, TJac* d_out, size_t ostride,
                     int nproofs, hipStream_t s);
// out[f * fstride + p * m + i] = sum_t sm[p * sstride + t m] * F_(t m + i) for the families f < nfam: T_b, U_b (per-proof 32-copy tables,
// row M | T_b | U_b, NP columns) and G_b (CRS copies d_ctab, nc columns, column map d_gb_cols)
void launch_late_uniform(const Fr* d_sm, size_t sstride, const TAff* d_ptab, size_t ptab_proof_stride, int NP, const TAff* d_ctab, int nc, const uint32_t* d_gb_cols, int n,
                         int m, int nfam, int split, TJac* d_out, size_t fstride, int nproofs, hipStream_t s);   // split: lanes per output (> 1 only while nfam * m * split <= 64); ceil(nfam * m * split / 64) waves per proof
void launch_late_tables(const TJac* d_jac, TAff* d_tab, TblTmp* d_tmp, int npoints, hipStream_t s);
void launch_late_msm(const LateRound& r, hipStream_t s);
void launch_late_restart(Fr* d_vec, int n, int m, int nproofs, hipStream_t s);

// ---- fused log rounds (round.hip): one launch per round of the IPA / SameMSM argument for small and mid-size batches ----
// grid = nproofs x wpp single-wave work-groups: per proof nf fixed-base tasks (16 / fix_wpw waves each) followed by nt table tasks
// (32 / tbl_wpw waves each), tasks in proof-major order.  Every wave reduces its own lane accumulators and leaves one XYZZ partial sum
// per set; the last wave of a proof to arrive adds them per request, finalises, hashes, draws the challenge and folds.
struct ProveDev;
struct RoundDev {
  const FixTask* ftasks;        // [nproofs][nf]
  const TblTask* ttasks;        // [nproofs][nt]
  int nf, nt, fix_wpw, tbl_wpw;
  uint32_t wpp;                 // waves per proof
  int nreq, nproofs;            // requests (cross terms) per proof: 4 (IPA) or 6 (SameMSM)
  uint32_t max_count;           // most partial sums of a request (<= 8)
  int next_scalars;             // 1: the tail also writes the next round's cross-term scalars to scal_out
  int combine;                  // 1 (SameMSM, fix_wpw 4, tbl_wpw 16): wave w of a proof runs fixed-base block w and table block w; wpp = nt * 2
  const TFix* fixtab;
  int nc;
  uint32_t *fraw, *fraw_slot, *traw, *traw_slot;   // raw sets of the fixed-base waves (one each) and of the table waves (two each)
  TAcc* part;                   // partial sums, indexed like the plan's partial slots
  const uint32_t* meta;         // [2][nproofs * nreq]: first partial | partial count of every request
  const uint32_t* comp_index;   // [nproofs * nreq]: slot-registry index of every request's 48-byte encoding
  uint32_t* counter;            // [nproofs] arrival counters, zero between launches
  Fr* scal_out;                 // the cross-term scalars of a round (IPA: [nproofs][2 n + 2], SameMSM: [nproofs][n])
};
bool round_fused_supported(int fix_wpw, int tbl_wpw, bool ipa);
void launch_round_fused(const RoundDev& rd, const ProveDev& d, int j, bool ipa, hipStream_t s);

void launch_sum_jac(const Jac* d_in, int n, Jac* d_out, int* d_flag, hipStream_t s);
void launch_bench_fpmul(Fp* d_data, int blocks, int iters, hipStream_t s);
void launch_bench_f28mul(Fp* d_data, int blocks, int iters, hipStream_t s);

}  // namespace cpx
