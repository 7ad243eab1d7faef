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
 * Checks: util::msm (util.rs:19-22) via cpx_g1_msm + cpx_g1_normalize against the expected encoding; CurdleproofsProof::new
 * (curdleproofs.rs:59) bytes against the expected proof; verify (curdleproofs.rs:197) accepts it and rejects it with two points
 * swapped; a per-context option round trip.  Exit code 0 and "cabi_smoke ok" on success. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "cpx.h"

typedef struct {
  uint8_t* p;
  uint64_t n;
} blob;

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
  uint8_t tmp[48];   /* swap A and cm_T.T_1: still valid encodings, a wrong proof */
  memcpy(tmp, proof, 48);
  memcpy(proof, proof + 48, 48);
  memcpy(proof + 48, tmp, 48);
  CHECK(cpx_batch_verify(ctx, proof, vrand.p, &verdict) == CPX_OK && verdict == CPX_ERR_VERIFY, "cpx_batch_verify must reject the mutated proof");
  cpx_ctx_destroy(ctx);
  printf("cabi_smoke ok: msm of %llu points, ell=%llu proof of %llu bytes bit-exact, verified, mutation rejected\n", (unsigned long long)(scalars.n / 32),
         (unsigned long long)ell, (unsigned long long)want_proof.n);
  return 0;
}
