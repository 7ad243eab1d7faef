/* cpx.h — C-ABI of the MI355X (gfx950) Curdleproofs G1 core.
 *
 * This is the drop-in boundary for the one hot path of asn-d6/curdleproofs: the BLS12-381 G1 group
 * arithmetic behind `CurdleproofsProof::{new,verify}`.  The reference is a pure-Rust crate with no
 * FFI of its own; the functions below are what an FFI for this path would bind, each one citing the
 * reference interface it replaces (paths relative to the reference repo, file:line).
 * INTEGRATION.md shows the Rust-side `extern "C"` block and the three edits that route the crate here.
 *
 * Data layouts — exactly arkworks' in-memory limbs, so a Rust caller passes slices through untouched:
 *   Fr     32 B   4 x u64 little-endian limbs, Montgomery form (R = 2^256)      = ark_bls12_381::Fr
 *   Fp     48 B   6 x u64 little-endian limbs, Montgomery form (R = 2^384)
 *   affine 96 B   x || y ; the point at infinity is encoded as 96 zero bytes     ~ G1Affine {x,y,infinity}
 *   jac   144 B   X || Y || Z Jacobian ; infinity <=> Z == 0                     = G1Projective
 *   compressed 48 B  zcash/ark-serialize compressed encoding
 *
 * Ownership: the caller owns every buffer it passes (valid for the duration of the call only); the
 * library owns device memory, the HIP stream and the device-resident CRS inside the opaque cpx_ctx.
 * Threading: one ctx = one HIP device + one stream; a ctx is not re-entrant, distinct ctxs may be driven
 * from distinct host threads.  Errors: int return, 0 = ok, negative = CPX_ERR_*; nothing unwinds
 * across this boundary (the reference's panics — util.rs:20, inner_product_argument.rs:115-116 —
 * become error codes).  There is NO CPU fallback: without a usable HIP device every call fails.
 */
#ifndef CPX_H
#define CPX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPX_OK 0
#define CPX_ERR_ARG (-1)         /* bad length / null pointer                                   */
#define CPX_ERR_NOT_POW2 (-2)    /* ell + 4 not a power of two (inner_product_argument.rs:116)  */
#define CPX_ERR_HIP (-3)         /* HIP runtime failure (no device, OOM, launch error)          */
#define CPX_ERR_VERIFY (-4)      /* ProofError::VerificationError (errors.rs:9)                 */
#define CPX_ERR_DESERIALIZE (-5) /* ark_serialize::SerializationError                           */
#define CPX_ERR_STATE (-6)       /* call order: CRS / batch not set                             */
#define CPX_ERR_INTERNAL (-7)

typedef struct cpx_ctx cpx_ctx;

/* ---- context ------------------------------------------------------------------------------- */
/* cpx_ctx_create fails with CPX_ERR_HIP when there is no such HIP device (there is no CPU fallback) and with CPX_ERR_STATE when the host
 * CPU lacks AVX2 / BMI2 / FMA / ADX (the library's host code is built for x86-64-v3 + ADX; every EPYC that hosts an MI355X has them) —
 * instead of dying of SIGILL later.  cpx_last_error(NULL) then says why the calling thread's last cpx_ctx_create failed.
 * The library does not touch the process environment.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and a context
 * owns four streams that are meant to overlap: export GPU_MAX_HW_QUEUES=8 before the process's first HIP call when two or more contexts
 * share a process (the Python package and bench.py do). */
int cpx_ctx_create(int device_id, cpx_ctx** out);
void cpx_ctx_destroy(cpx_ctx* ctx);
const char* cpx_last_error(const cpx_ctx* ctx); /* message of the last failing call on this ctx    */
int cpx_device_count(void);

/* Per-context tunables (kernel selection thresholds, radix of the fixed-base table, batch size from which the protocol runs on the
 * device, ...): key = one of the names listed in curdleproofs_amd/csrc/kernels.h `struct Options` ("fix_bits", "device_min_batch",
 * "tbw_wpw", "reduce_wave_max", ...).  A new context starts from the built-in defaults, overridden by CPX_<KEY> environment
 * variables read once per process; everything after that goes through these calls and concerns this context only.  Results never
 * depend on an option (tests/test_gpu_parity.py::test_engine_variants_stay_bit_exact), only speed does.  "fix_bits" takes effect at
 * the next cpx_ctx_set_crs.  Unknown key / value out of range -> CPX_ERR_ARG.  cpx_ctx_get_option also answers the read-only key
 * "fix_bits_effective": the radix of the fixed-base table this context actually uses (the fallback of cpx_ctx_set_crs; 0 = no CRS). */
int cpx_ctx_set_option(cpx_ctx* ctx, const char* key, long long value);
int cpx_ctx_get_option(const cpx_ctx* ctx, const char* key, long long* value);

/* Optional: page-locked host buffers.  Any host memory may be passed to the calls below; buffers obtained here let the batch calls
 * move their inputs / outputs by asynchronous DMA (no staging copies, transfers overlap the kernels of other contexts). */
void* cpx_host_alloc(size_t bytes);
void cpx_host_free(void* p);

/* crs.rs:37-58 `CurdleproofsCrs::from_points(ell, points)`: ell + 7 affine points in the order
 * vec_G[ell] | vec_H[4] | H | G_t | G_u.  Uploads them, computes G_sum / H_sum on the device and builds the fixed-base
 * tables (option "fix_bits": radix 2^16 by default, 17.5 GB at ell = 252; 2^19 = 122 GB; falls back 19 -> 16 -> 8 when free HBM is short).  `n_points` is the number of
 * points behind `points`: fewer than ell + 7 is the reference's "not enough points" error (crs.rs:40-42) ->
 * CPX_ERR_ARG; surplus points are ignored, as the reference's slicing does.  On failure the context holds no CRS. */
int cpx_ctx_set_crs(cpx_ctx* ctx, size_t ell, const uint8_t* points, size_t n_points);
int cpx_crs_sums(const cpx_ctx* ctx, uint8_t g_sum[96], uint8_t h_sum[96]);
size_t cpx_proof_size(const cpx_ctx* ctx); /* 48*(18+10*log2(ell+4)) + 32*7, e.g. 4928 at ell = 252 */
size_t cpx_batch_size(const cpx_ctx* ctx); /* instances currently loaded (cpx_batch_load, or 1 after a cpx_whisk_*_shuffle_proof call); 0 = none */

/* ---- tier 0: the reference's MSM funnel and the loops that bypass it ------------------------- */
/* util.rs:19-22 `msm(points: &[G1Affine], scalars: &[Fr]) -> G1Projective` */
int cpx_g1_msm(cpx_ctx* ctx, const uint8_t* bases /* n*96 */, const uint8_t* scalars /* n*32 */, size_t n, uint8_t out[144]);
/* util.rs:25-29 `msm_from_projective(points: &[G1Projective], scalars: &[Fr])` */
int cpx_g1_msm_jac(cpx_ctx* ctx, const uint8_t* bases /* n*144 */, const uint8_t* scalars, size_t n, uint8_t out[144]);
/* inner_product_argument.rs:177-178, same_multiscalar_argument.rs:128-130:
 *   PL[i] <- (PL[i] + PR[i] * gamma).into_affine(), i < half, in place */
int cpx_g1_fold(cpx_ctx* ctx, uint8_t* PL /* half*96 */, const uint8_t* PR /* half*96 */, const uint8_t gamma[32], size_t half);
/* grand_product_argument.rs:90-102 (per-element scalars, scalar_stride = 32) and util.rs:94-95
 * (one shared scalar, scalar_stride = 0):  out[i] <- (P[i] * s_i).into_affine()
 * PRECONDITION of cpx_g1_fold / cpx_g1_scale (as of arkworks' G1Affine, whose deserialisation checks it): the points lie in the order-r
 * subgroup — the scalar is split by the G1 endomorphism, k P = t P + q (-phi(P)), an identity of that subgroup only; on a point of
 * E(Fp) outside it a scalar >= z^2 / 2 gives a wrong result.  Context option "scale_any_point" = 1 selects the plain 257-step
 * double-and-add for both calls, valid on all of E(Fp) (the cofactor multiplication of the hash-to-curve CRS, tests/crs.rs:38). */
int cpx_g1_scale(cpx_ctx* ctx, const uint8_t* P /* n*96 */, const uint8_t* scalars, size_t scalar_stride, size_t n, uint8_t* out /* n*96 */);
/* ark_ec `CurveGroup::normalize_batch` (+ optional `serialize_compressed`); either output may be NULL */
int cpx_g1_normalize(cpx_ctx* ctx, const uint8_t* jac /* n*144 */, size_t n, uint8_t* out_affine /* n*96 */, uint8_t* out_compressed /* n*48 */);
/* whisk.rs:318-320 `from_bytes_g1affine` = deserialize_compressed with on-curve + subgroup validation */
int cpx_g1_decompress(cpx_ctx* ctx, const uint8_t* compressed /* n*48 */, size_t n, uint8_t* out_affine /* n*96 */, int check_subgroup);

/* The same with one verdict per point instead of one for the call: status[i] = 0 ok (out_affine[i] valid), 1 malformed encoding or x
 * not on the curve, 2 on the curve but outside the order-r subgroup.  Returns CPX_OK whatever the verdicts are.  This is the
 * device square root behind a try-and-increment hash to the curve: tests/crs.rs:13-52 `generate_random_points` maps a hash to x and
 * keeps `get_point_from_x_unchecked(x, false)` when it exists (curdleproofs_amd/crs.py::generate_random_points). */
int cpx_g1_decompress_status(cpx_ctx* ctx, const uint8_t* compressed /* n*48 */, size_t n, uint8_t* out_affine /* n*96 */, int check_subgroup, uint8_t* status /* n */);

/* ---- tier 1: msm_accumulator.rs:22-68 `MsmAccumulator` -------------------------------------- */
typedef struct cpx_accum cpx_accum;
int cpx_accum_new(cpx_ctx* ctx, cpx_accum** out);
void cpx_accum_free(cpx_accum* acc);
/* `accumulate_check(&mut self, C, vec_x, vec_V, rng)`; the caller passes the `Fr::rand(rng)` value it drew: a uniform,
 * reduced, NON-ZERO field element (zero or >= r -> CPX_ERR_ARG: a zero factor would drop the check) */
int cpx_accum_check(cpx_accum* acc, const uint8_t C[144], const uint8_t* vec_x /* n*32 */, const uint8_t* vec_V /* n*96 */, size_t n,
                    const uint8_t random_factor[32]);
/* `verify(self)`: CPX_OK or CPX_ERR_VERIFY */
int cpx_accum_verify(cpx_accum* acc);

/* ---- tier 2: whole proofs, batched, instance data resident in HBM --------------------------- */
/* Uploads `batch` instances (vec_R, vec_S, vec_T, vec_U: batch*ell*96 each; M: batch*144 Jacobian):
 * the public inputs of curdleproofs.rs:59-70 / :197-207. */
int cpx_batch_load(cpx_ctx* ctx, size_t batch, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T, const uint8_t* vec_U,
                   const uint8_t* M);
/* The same upload in two halves, for a caller that always has a next batch (a service draining a queue of shuffles): _begin starts moving the
 * NEXT batch's instance to a staging area in HBM on an upload stream of its own and returns at once — the batch loaded before stays loaded, and
 * cpx_batch_prove / cpx_batch_verify on it run beside the transfer; _end waits for the transfer (the host buffers, which must stay untouched
 * from _begin to _end, are free again when it returns) and makes the staged batch the loaded one (device-to-device, behind the context's last
 * kernel).  97 KB per proof at ell = 252 cross PCIe per instance: with cpx_batch_load in every pass the bench rate is 5 % lower, with
 * _begin before the prove and _end after the verify it is within 1 % (bench.py `value_incl_instance_upload`).  Page-locked buffers
 * (cpx_host_alloc) make the transfer asynchronous; pageable ones work, the copy then happens inside _begin.
 * _end without _begin, or a cpx_ctx_set_crs in between -> CPX_ERR_STATE. */
int cpx_batch_load_begin(cpx_ctx* ctx, size_t batch, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T, const uint8_t* vec_U,
                         const uint8_t* M);
int cpx_batch_load_end(cpx_ctx* ctx);
/* curdleproofs.rs:59 `CurdleproofsProof::new` for every loaded instance.
 *   permutation  batch*ell u32          k  batch*32          vec_m_blinders  batch*4*32
 *   rand         batch*(3n+9)*32, n = ell+4: the `Fr::rand(rng)` draws in the reference's order —
 *                vec_a_blinders[2] (curdleproofs.rs:86), vec_c_blinders[4] (grand_product_argument.rs:75),
 *                IPA r[n], z[n-2] (inner_product_argument.rs:46-47), r_t, r_u (curdleproofs.rs:110-111),
 *                r_a, r_b, r_k (same_scalar_argument.rs:56-58), vec_r[n] (same_multiscalar_argument.rs:78)
 *   proofs_out   batch*cpx_proof_size() bytes = `CurdleproofsProof::serialize` (curdleproofs.rs:300-310) */
int cpx_batch_prove(cpx_ctx* ctx, const uint32_t* permutation, const uint8_t* k, const uint8_t* vec_m_blinders, const uint8_t* rand,
                    uint8_t* proofs_out);
/* curdleproofs.rs:197 `verify` (after `deserialize`, :312-323) for every loaded instance.
 *   rand     batch*8*32: the eight `accumulate_check` factors (msm_accumulator.rs:44) in call order.  They must be
 *            independent uniform non-zero reduced field elements (what `Fr::rand` returns but for the zero it may
 *            return with probability 1/r); zero or >= r -> CPX_ERR_ARG.
 *   verdict  batch ints: CPX_OK, CPX_ERR_VERIFY or CPX_ERR_DESERIALIZE
 * One deliberate difference in HOW (not in what) is checked: the reference tests the four SameScalar relations
 * (same_scalar_argument.rs:127-137) as direct equalities; here they join the accumulated MSM with the weights
 * a1*a2, a3*a4, a5*a6, a7*a8 built from the caller's factors.  The accumulated sum is then a polynomial of total degree 2
 * in independent uniform factors whose coefficients are the individual check values, so a false relation survives with
 * probability <= 2/r (Schwartz-Zippel) — the same bound every other check of `MsmAccumulator` already rests on.  The
 * verdict therefore equals the reference's except with probability < 2^-253 over the caller's factors
 * (tests/test_gpu_parity.py::test_same_scalar_relations_are_checked isolates these relations).
 * Point encodings with the INFINITY FLAG set (context option "strict_infinity"): ark-bls12-381 ^0.4's `read_g1_compressed` — behind
 * `G1Affine::deserialize_compressed`, curdleproofs.rs:312-323, whisk.rs:313-320 — returns the identity as soon as the compression and
 * the infinity flag are set and looks at neither the sort flag nor the remaining 381 bits (ark-bls12-381 0.4.0 `curves/util.rs`, recalled:
 * the crate is not in the build image; version 0.5 rejects both).  strict_infinity = 0 (default) reproduces that: such an encoding IS the
 * identity — and is hashed in its canonical form 0xc0 || 0^47, as the reference hashes the deserialised point — so a proof carrying one
 * gets the verdict the reference gives it (normally CPX_ERR_VERIFY, not CPX_ERR_DESERIALIZE).  strict_infinity = 1: only 0xc0 || 0^47
 * is the identity, anything else with the flag set is CPX_ERR_DESERIALIZE (the zcash specification's wording).  The option also governs
 * cpx_g1_decompress / cpx_g1_decompress_status and the cpx_whisk_* entry points. */
int cpx_batch_verify(cpx_ctx* ctx, const uint8_t* proofs, const uint8_t* rand, int* verdict);
/* BASELINE config 5: the batched verifier.  Every `accumulate_check` of every loaded proof goes into
 * ONE accumulated MSM — the reference's `MsmAccumulator` (msm_accumulator.rs:22-68; `new` is pub(crate), :28, and
 * `CurdleproofsProof::verify` makes one per call, curdleproofs.rs:215) shared by all verify calls; the SameScalar
 * equalities (same_scalar_argument.rs:127-137) join it with their own random weights.
 *   rand         batch*12*32: per proof the eight `accumulate_check` factors, then four weights for the equalities
 *   partial_jac  144 B: this context's share  sum (lhs - rhs)  of the accumulated check (Jacobian, standard form).
 *                The batch is accepted iff the partial sums of all contexts / GPUs add up to the identity
 *                (all-gather the 144-byte partials, add them with cpx_g1_sum_jac) and no context reports invalid proofs.
 *   n_invalid    number of loaded proofs that failed deserialisation or the structural checks (curdleproofs.rs:218)
 * All-or-nothing: a rejected batch does not say which proof is wrong (re-run cpx_batch_verify for that). */
int cpx_batch_verify_fused(cpx_ctx* ctx, const uint8_t* proofs, const uint8_t* rand, uint8_t* partial_jac, int* n_invalid);
/* sum of n Jacobian points (n*144 B) -> out (144 B); *is_identity = 1 iff the sum is the point at infinity */
int cpx_g1_sum_jac(cpx_ctx* ctx, const uint8_t* points_jac, size_t n, uint8_t* out_jac, int* is_identity);

/* ---- Whisk byte-level API: whisk.rs (the reference's `pub fn`s, source-compatible at the byte level) ------------------
 * A tracker is 96 bytes: r_G (48, compressed) || k_r_G (48)  (`WhiskTracker`, whisk.rs:36-42).  ell is the CRS's ell
 * (the reference hard-codes N = 128, ell = 124, whisk.rs:27-28; any power-of-two ell + 4 works here).  A bad point or
 * scalar encoding anywhere is the reference's `Err(SerializationError)` -> CPX_ERR_DESERIALIZE.  The RNG stays with the
 * caller: every draw the reference makes from `rng` is an argument, in the reference's order. */
/* whisk.rs:144-179 `generate_whisk_shuffle_proof(rng, crs, pre_trackers)`.
 *   permutation     ell u32: `(0..ELL).collect().shuffle(rng)` (:152-155)      k  `Fr::rand(rng)` (:156)
 *   vec_m_blinders  4*32: `generate_blinders(rng, N_BLINDERS)` inside shuffle_permute_and_commit_input (util.rs:91)
 *   rand            (3n+9)*32: the draws of `CurdleproofsProof::new` (see cpx_batch_prove)
 *   post_trackers_out  ell*96       proof_out  48 + cpx_proof_size() = M || CurdleproofsProof::serialize (4496 B at ell = 124)
 * Replaces the batch loaded into ctx (it loads the one instance it proves). */
int cpx_whisk_generate_shuffle_proof(cpx_ctx* ctx, const uint8_t* pre_trackers /* ell*96 */, const uint32_t* permutation, const uint8_t k[32],
                                     const uint8_t* vec_m_blinders, const uint8_t* rand, uint8_t* post_trackers_out, uint8_t* proof_out);
/* whisk.rs:106-130 `is_valid_whisk_shuffle_proof(rng, crs, pre_trackers, post_trackers, proof_bytes)`;
 *   rand 8*32: the verifier's `accumulate_check` factors (see cpx_batch_verify).  *valid = 1 / 0 = Ok(true) / Ok(false). */
int cpx_whisk_is_valid_shuffle_proof(cpx_ctx* ctx, const uint8_t* pre_trackers, const uint8_t* post_trackers, const uint8_t* proof, const uint8_t* rand,
                                     int* valid);
/* whisk.rs:228-263 `generate_whisk_tracker_proof(rng, tracker, k)`; blinder = its one `Fr::rand(rng)` (:238).
 *   proof_out 128 B = A (48) || B (48) || s (32 LE)  (`TrackerProof`, whisk.rs:69-73) */
int cpx_whisk_generate_tracker_proof(cpx_ctx* ctx, const uint8_t tracker[96], const uint8_t k[32], const uint8_t blinder[32], uint8_t proof_out[128]);
/* whisk.rs:183-226 `is_valid_whisk_tracker_proof(tracker, k_commitment, tracker_proof)` */
int cpx_whisk_is_valid_tracker_proof(cpx_ctx* ctx, const uint8_t tracker[96], const uint8_t k_commitment[48], const uint8_t proof[128], int* valid);

/* ---- measurement --------------------------------------------------------------------------- */
int cpx_set_profiling(cpx_ctx* ctx, int on); /* time every kernel group with HIP events on the ctx stream */
int cpx_reset_stats(cpx_ctx* ctx);
/* name = kernel name as rocprofv3 reports it, template arguments included: "k_msm_fix<16, 16>", "k_msm_tblw<32, false>",
 * "k_msm_tblw<2, true>", "k_msm_accw", "k_reduce_sets", "k_finalize_ranges", "k_table_build", "k_msm_tail", "k_smul",
 * "k_finalize", "k_compress", "k_decompress" (+ the host spans "host_prove_wall", "host_verify_wall", ...);
 * units = MSM points / scalar-mul elements / points; out pointers may be NULL */
int cpx_get_stat(const cpx_ctx* ctx, const char* name, uint64_t* launches, double* total_ms, double* algorithmic_bytes, double* units);
int cpx_set_host_threads(cpx_ctx* ctx, int threads);
/* dependent Fp multiply chains: returns achieved Fp products per second */
int cpx_bench_fpmul(cpx_ctx* ctx, int blocks, int iters, int reps, double* products_per_second);

#ifdef __cplusplus
}
#endif
#endif /* CPX_H */
