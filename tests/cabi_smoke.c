/* cabi_smoke.c — a plain C caller of the drop-in boundary (include/cpx.h), compiled with gcc against the header and linked
 * against libcpx.so; no Python, no ctypes in the call path.  tests/test_cabi_c_caller.py builds it, writes the input file with
 * the CPU oracle (test infrastructure) and runs it on the GPU box.
 *
 *   cabi_smoke <input file>
 *
 * Input file: a sequence of blobs, each a u64 little-endian length followed by that many bytes, in this order:
 *   ell (8 bytes) | crs points (ell+7)*96 | msm bases n*96 | msm scalars n*32 | expected msm result, compressed 48 |
 *   vec_R | vec_S | vec_T | vec_U (ell*96 each) | M 144 | permutation ell*4 | k 32 | vec_m_blinders 128 | prover rand (3n+9)*32 |
 *   expected proof | verifier rand 8*32
 * then (optional second section, present when more blobs follow):
 *   accumulator: V n*96 | x n*32 | C = msm(V, x) 144 | C' != C 144 | two random factors 64 |
 *   fused verifier: 12 random factors 12*32 |
 *   Whisk tracker KAT (whisk.rs:381-402): tracker 96 | k 32 | blinder 32 | k_commitment 48 | expected 128-byte proof |
 *   Whisk shuffle KAT (whisk.rs:416-456): ell2 8 | crs2 (ell2+7)*96 | pre-trackers ell2*96 | permutation ell2*4 | k 32 | vec_m_blinders 128 |
 *                                         prover rand (3 n2 + 9)*32 | expected M || proof (the reference's 4496-byte vector) | verifier rand 8*32
 * Checks: util::msm (util.rs:19-22) via cpx_g1_msm + cpx_g1_normalize against the expected encoding; CurdleproofsProof::new
 * (curdleproofs.rs:59) bytes against the expected proof; verify (curdleproofs.rs:197) accepts it and rejects it with two points
 * swapped; a per-context option round trip.  Second section: MsmAccumulator (msm_accumulator.rs:22-68) through cpx_accum_* — two true
 * checks over shared bases verify, a false one is CPX_ERR_VERIFY, a zero factor is refused; BASELINE configs[4] through
 * cpx_batch_verify_fused + cpx_g1_sum_jac on the ell = 28 proof (identity for the valid proof, not for the mutated one); the four
 * cpx_whisk_* entry points (whisk.rs:106-263) against the reference's own 128-byte and 4496-byte vectors, and Ok(false) on a
 * tampered proof.  Exit code 0 and "cabi_smoke ok" on success. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "cpx.h"

typedef struct {
  uint8_t* p;
  uint64_t n;
} blob;

static int more_blobs(FILE* f) {
  int c = fgetc(f);
  if (c == EOF) return 0;
  ungetc(c, f);
  return 1;
}

static blob read_blob(FILE* f) {
  blob b = {NULL, 0};
  if (fread(&b.n, 8, 1, f) != 1) {
    fprintf(stderr, "input file truncated\n");
    exit(2);
  }
  b.p = (uint8_t*)malloc(b.n ? b.n : 1);
  if (b.n && fread(b.p, 1, b.n, f) != b.n) {
    fprintf(stderr, "input file truncated\n");
    exit(2);
  }
  return b;
}

#define CHECK(cond, what)                                                                   \
  do {                                                                                      \
    if (!(cond)) {                                                                          \
      fprintf(stderr, "cabi_smoke FAILED: %s (%s)\n", what, ctx ? cpx_last_error(ctx) : ""); \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)

int main(int argc, char** argv) {
  cpx_ctx* ctx = NULL;
  if (argc < 2) {
    fprintf(stderr, "usage: cabi_smoke <input file>\n");
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  blob ellb = read_blob(f), crs = read_blob(f), bases = read_blob(f), scalars = read_blob(f), want_msm = read_blob(f);
  blob R = read_blob(f), S = read_blob(f), T = read_blob(f), U = read_blob(f), M = read_blob(f);
  blob perm = read_blob(f), k = read_blob(f), mbl = read_blob(f), prand = read_blob(f), want_proof = read_blob(f), vrand = read_blob(f);
  const int second = more_blobs(f);
  blob aV = {0}, ax = {0}, aC = {0}, aCbad = {0}, afac = {0}, frand = {0}, trk = {0}, tk = {0}, tbl = {0}, tkc = {0}, tproof = {0};
  blob ell2b = {0}, crs2 = {0}, pre = {0}, perm2 = {0}, k2 = {0}, mbl2 = {0}, prand2 = {0}, want_shuffle = {0}, vrand2 = {0};
  if (second) {
    aV = read_blob(f), ax = read_blob(f), aC = read_blob(f), aCbad = read_blob(f), afac = read_blob(f), frand = read_blob(f);
    trk = read_blob(f), tk = read_blob(f), tbl = read_blob(f), tkc = read_blob(f), tproof = read_blob(f);
    ell2b = read_blob(f), crs2 = read_blob(f), pre = read_blob(f), perm2 = read_blob(f), k2 = read_blob(f), mbl2 = read_blob(f), prand2 = read_blob(f);
    want_shuffle = read_blob(f), vrand2 = read_blob(f);
  }
  fclose(f);
  uint64_t ell = 0;
  memcpy(&ell, ellb.p, 8);

  CHECK(cpx_device_count() >= 1, "no HIP device: the library has no CPU fallback");
  CHECK(cpx_ctx_create(0, &ctx) == CPX_OK, "cpx_ctx_create");
  long long v = 0;
  CHECK(cpx_ctx_set_option(ctx, "tail_wave_max", 128) == CPX_OK && cpx_ctx_get_option(ctx, "tail_wave_max", &v) == CPX_OK && v == 128, "option round trip");
  CHECK(cpx_ctx_set_option(ctx, "no_such_option", 1) == CPX_ERR_ARG, "unknown option must be refused");
  CHECK(cpx_ctx_set_crs(ctx, (size_t)ell, crs.p, (size_t)(crs.n / 96)) == CPX_OK, "cpx_ctx_set_crs");
  CHECK(cpx_proof_size(ctx) == want_proof.n, "cpx_proof_size");

  /* util::msm */
  uint8_t jac[144], comp[48];
  CHECK(cpx_g1_msm(ctx, bases.p, scalars.p, (size_t)(scalars.n / 32), jac) == CPX_OK, "cpx_g1_msm");
  CHECK(cpx_g1_normalize(ctx, jac, 1, NULL, comp) == CPX_OK, "cpx_g1_normalize");
  CHECK(want_msm.n == 48 && !memcmp(comp, want_msm.p, 48), "msm result differs from the oracle's");

  /* CurdleproofsProof::new / verify */
  CHECK(cpx_batch_load(ctx, 1, R.p, S.p, T.p, U.p, M.p) == CPX_OK && cpx_batch_size(ctx) == 1, "cpx_batch_load");
  uint8_t* proof = (uint8_t*)malloc(want_proof.n);
  CHECK(cpx_batch_prove(ctx, (const uint32_t*)perm.p, k.p, mbl.p, prand.p, proof) == CPX_OK, "cpx_batch_prove");
  CHECK(!memcmp(proof, want_proof.p, want_proof.n), "proof bytes differ from the oracle's");
  int verdict = CPX_ERR_INTERNAL;
  CHECK(cpx_batch_verify(ctx, proof, vrand.p, &verdict) == CPX_OK && verdict == CPX_OK, "cpx_batch_verify must accept the proof");
  /* the same instance staged as the NEXT batch (cpx_batch_load_begin / _end): the loaded batch keeps proving in between, the staged one after */
  CHECK(cpx_batch_load_end(ctx) == CPX_ERR_STATE, "cpx_batch_load_end without _begin must be a state error");
  CHECK(cpx_batch_load_begin(ctx, 1, R.p, S.p, T.p, U.p, M.p) == CPX_OK, "cpx_batch_load_begin");
  CHECK(cpx_batch_prove(ctx, (const uint32_t*)perm.p, k.p, mbl.p, prand.p, proof) == CPX_OK && !memcmp(proof, want_proof.p, want_proof.n), "prove beside a staged upload");
  CHECK(cpx_batch_load_end(ctx) == CPX_OK && cpx_batch_size(ctx) == 1, "cpx_batch_load_end");
  CHECK(cpx_batch_prove(ctx, (const uint32_t*)perm.p, k.p, mbl.p, prand.p, proof) == CPX_OK && !memcmp(proof, want_proof.p, want_proof.n), "prove on the staged batch");
  uint8_t tmp[48];   /* swap A and cm_T.T_1: still valid encodings, a wrong proof */
  memcpy(tmp, proof, 48);
  memcpy(proof, proof + 48, 48);
  memcpy(proof + 48, tmp, 48);
  CHECK(cpx_batch_verify(ctx, proof, vrand.p, &verdict) == CPX_OK && verdict == CPX_ERR_VERIFY, "cpx_batch_verify must reject the mutated proof");
  if (second) {
    /* BASELINE configs[4]: one accumulated check (here over the one loaded proof), the partial sum judged by cpx_g1_sum_jac */
    uint8_t part[144], sum[144];
    int n_invalid = -1, is_id = -1;
    CHECK(cpx_batch_verify_fused(ctx, proof, frand.p, part, &n_invalid) == CPX_OK && n_invalid == 0, "cpx_batch_verify_fused (mutated proof)");
    CHECK(cpx_g1_sum_jac(ctx, part, 1, sum, &is_id) == CPX_OK && is_id == 0, "the partial sum of a wrong proof must not be the identity");
    CHECK(cpx_batch_verify_fused(ctx, want_proof.p, frand.p, part, &n_invalid) == CPX_OK && n_invalid == 0, "cpx_batch_verify_fused");
    CHECK(cpx_g1_sum_jac(ctx, part, 1, sum, &is_id) == CPX_OK && is_id == 1, "the partial sum of a valid proof must be the identity");
    uint8_t two[288];   /* P + (the same partial again) stays the identity; the all-gather step adds the partials of all GPUs like this */
    memcpy(two, part, 144);
    memcpy(two + 144, part, 144);
    CHECK(cpx_g1_sum_jac(ctx, two, 2, sum, &is_id) == CPX_OK && is_id == 1, "cpx_g1_sum_jac over two partials");

    /* MsmAccumulator (msm_accumulator.rs:22-68) */
    cpx_accum* acc = NULL;
    const size_t an = (size_t)(ax.n / 32);
    CHECK(cpx_accum_new(ctx, &acc) == CPX_OK, "cpx_accum_new");
    CHECK(cpx_accum_check(acc, aC.p, ax.p, aV.p, an, afac.p) == CPX_OK, "cpx_accum_check");
    CHECK(cpx_accum_check(acc, aC.p, ax.p, aV.p, an, afac.p + 32) == CPX_OK, "cpx_accum_check (same bases again: the scalars merge)");
    CHECK(cpx_accum_verify(acc) == CPX_OK, "cpx_accum_verify must accept two true relations");
    cpx_accum_free(acc);
    CHECK(cpx_accum_new(ctx, &acc) == CPX_OK, "cpx_accum_new");
    CHECK(cpx_accum_check(acc, aC.p, ax.p, aV.p, an, afac.p) == CPX_OK, "cpx_accum_check");
    CHECK(cpx_accum_check(acc, aCbad.p, ax.p, aV.p, an, afac.p + 32) == CPX_OK, "cpx_accum_check (false relation)");
    CHECK(cpx_accum_verify(acc) == CPX_ERR_VERIFY, "cpx_accum_verify must reject a false relation");
    uint8_t zero[32] = {0};
    CHECK(cpx_accum_check(acc, aC.p, ax.p, aV.p, an, zero) == CPX_ERR_ARG, "a zero random factor must be refused");
    cpx_accum_free(acc);

    /* Whisk: tracker proof (whisk.rs:228-263, :183-226) against the reference's 128-byte vector */
    uint8_t tp[128];
    int valid = -1;
    CHECK(tproof.n == 128 && cpx_whisk_generate_tracker_proof(ctx, trk.p, tk.p, tbl.p, tp) == CPX_OK, "cpx_whisk_generate_tracker_proof");
    CHECK(!memcmp(tp, tproof.p, 128), "tracker proof differs from the reference's vector (whisk.rs:401)");
    CHECK(cpx_whisk_is_valid_tracker_proof(ctx, trk.p, tkc.p, tp, &valid) == CPX_OK && valid == 1, "cpx_whisk_is_valid_tracker_proof");
    tp[100] ^= 1;   /* the response s changed: Ok(false) */
    CHECK(cpx_whisk_is_valid_tracker_proof(ctx, trk.p, tkc.p, tp, &valid) == CPX_OK && valid == 0, "a tampered tracker proof must be Ok(false)");
    memset(tp + 96, 0xff, 32);   /* s >= r: Err(SerializationError) */
    CHECK(cpx_whisk_is_valid_tracker_proof(ctx, trk.p, tkc.p, tp, &valid) == CPX_ERR_DESERIALIZE, "a non-canonical scalar must be a deserialisation error");

    /* Whisk: shuffle proof (whisk.rs:144-179, :106-130) against the reference's 4496-byte vector */
    uint64_t ell2 = 0;
    memcpy(&ell2, ell2b.p, 8);
    CHECK(cpx_ctx_set_crs(ctx, (size_t)ell2, crs2.p, (size_t)(crs2.n / 96)) == CPX_OK, "cpx_ctx_set_crs (ell = 124)");
    CHECK(want_shuffle.n == 48 + cpx_proof_size(ctx), "shuffle proof size");
    uint8_t* post = (uint8_t*)malloc(pre.n);
    uint8_t* sp = (uint8_t*)malloc(want_shuffle.n);
    CHECK(cpx_whisk_generate_shuffle_proof(ctx, pre.p, (const uint32_t*)perm2.p, k2.p, mbl2.p, prand2.p, post, sp) == CPX_OK, "cpx_whisk_generate_shuffle_proof");
    CHECK(!memcmp(sp, want_shuffle.p, want_shuffle.n), "shuffle proof differs from the reference's vector (whisk.rs:455)");
    CHECK(cpx_batch_size(ctx) == 1, "the shuffle call loads its one instance");
    CHECK(cpx_whisk_is_valid_shuffle_proof(ctx, pre.p, post, sp, vrand2.p, &valid) == CPX_OK && valid == 1, "cpx_whisk_is_valid_shuffle_proof");
    CHECK(cpx_whisk_is_valid_shuffle_proof(ctx, post, pre.p, sp, vrand2.p, &valid) == CPX_OK && valid == 0, "pre / post exchanged must be Ok(false)");
    sp[0] &= 0x7f;   /* M without the compression flag: Err(SerializationError) */
    CHECK(cpx_whisk_is_valid_shuffle_proof(ctx, pre.p, post, sp, vrand2.p, &valid) == CPX_ERR_DESERIALIZE, "an undecodable M must be a deserialisation error");
    printf("cabi_smoke second section ok: accumulator, fused verifier, Whisk tracker proof (128 B) and shuffle proof (%llu B) equal to the reference's vectors\n",
           (unsigned long long)want_shuffle.n);
  }
  cpx_ctx_destroy(ctx);
  printf("cabi_smoke ok: msm of %llu points, ell=%llu proof of %llu bytes bit-exact, verified, mutation rejected\n", (unsigned long long)(scalars.n / 32),
         (unsigned long long)ell, (unsigned long long)want_proof.n);
  return 0;
}
