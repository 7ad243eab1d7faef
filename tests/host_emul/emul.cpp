// TEST-ONLY: compiles the product's host/device-portable math (curdleproofs_amd/csrc/*.cuh,
// host_math.hpp) for the CPU so that `pytest -m "not gpu"` can check the exact limb arithmetic,
// point formulas, scalar recodings and the engine's transcript against the oracle without a GPU.
// This is NOT a product path: nothing in curdleproofs_amd/ links or loads it.
#include <cstring>
#include <vector>
#include "../../curdleproofs_amd/csrc/g1.cuh"
#include "../../curdleproofs_amd/csrc/recode.cuh"
#include "../../curdleproofs_amd/csrc/host_math.hpp"

using namespace cpx;

extern "C" {

void emul_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Fp x, y;
    memcpy(x.v, a + 48 * i, 48);
    memcpy(y.v, b + 48 * i, 48);
    Fp z = fe_mul(x, y);
    memcpy(out + 48 * i, z.v, 48);
  }
}
void emul_fp_addsubnegdbl(const uint8_t* a, const uint8_t* b, uint8_t* out /*4*48*/) {
  Fp x, y;
  memcpy(x.v, a, 48);
  memcpy(y.v, b, 48);
  Fp r[4] = {fe_add(x, y), fe_sub(x, y), fe_neg(x), fe_dbl(x)};
  memcpy(out, r, sizeof r);
}
void emul_fp_inv(const uint8_t* a, uint8_t* out) {
  Fp x;
  memcpy(x.v, a, 48);
  Fp z = fe_inv(x);
  memcpy(out, z.v, 48);
}
void emul_fr_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Fr x, y;
    memcpy(x.v, a + 32 * i, 32);
    memcpy(y.v, b + 32 * i, 32);
    Fr z = fe_mul(x, y);
    memcpy(out + 32 * i, z.v, 32);
  }
}
void emul_fr_inv(const uint8_t* a, uint8_t* out) {
  host::S s;
  memcpy(s.f.v, a, 32);
  s = s.inverse();
  memcpy(out, s.f.v, 32);
}

// the MSM exactly as k_msm_accum + k_msm_tail(8,4) + k_msm_tail(8,32) compute it
void emul_msm(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t* out) {
  const Aff* P = reinterpret_cast<const Aff*>(bases);
  std::vector<Jac> bucket(512, Jac::identity());
  std::vector<int8_t> dig(64);
  for (size_t i = 0; i < n; i++) {
    Fr k;
    memcpy(k.v, scalars + 32 * i, 32);
    k = fe_from_mont(k);
    recode_signed16(k.v, dig.data());
    for (int w = 0; w < 64; w++) {
      int d = dig[w];
      if (!d) continue;
      int mag = d < 0 ? -d : d;
      Jac& b = bucket[w * 8 + mag - 1];
      b = jac_add_mixed(b, aff_cneg(P[i], d < 0));
    }
  }
  Jac wsum[64];
  for (int w = 0; w < 64; w++) {
    Jac run = Jac::identity(), tot = Jac::identity();
    for (int b = 7; b >= 0; b--) {
      run = jac_add(run, bucket[w * 8 + b]);
      tot = jac_add(tot, run);
    }
    wsum[w] = tot;
  }
  auto tail = [](const Jac* in, Jac* o, int nout, int group, int shift) {
    for (int t = 0; t < nout; t++) {
      const Jac* src = in + t * group;
      Jac acc = src[group - 1];
      for (int j = group - 2; j >= 0; j--) {
        for (int s = 0; s < shift; s++) acc = jac_dbl(acc);
        acc = jac_add(acc, src[j]);
      }
      o[t] = acc;
    }
  };
  Jac part[8], res;
  tail(wsum, part, 8, 8, 4);
  tail(part, &res, 1, 8, 32);
  memcpy(out, &res, sizeof res);
}

// k_smul's per-element computation: (add ? add : O) + s*mul via NAF, Jacobian out
void emul_smul(const uint8_t* add, const uint8_t* mul, const uint8_t* scalar, uint8_t* out) {
  Fr k;
  memcpy(k.v, scalar, 32);
  k = fe_from_mont(k);
  uint32_t nz[9], ng[9];
  recode_naf(k.v, nz, ng);
  Aff P;
  memcpy(&P, mul, sizeof P);
  Jac acc = Jac::identity();
  for (int i = 256; i >= 0; i--) {
    acc = jac_dbl(acc);
    if ((nz[i >> 5] >> (i & 31)) & 1u) acc = jac_add_mixed(acc, aff_cneg(P, (ng[i >> 5] >> (i & 31)) & 1u));
  }
  if (add) {
    Aff A;
    memcpy(&A, add, sizeof A);
    acc = jac_add_mixed(acc, A);
  }
  memcpy(out, &acc, sizeof acc);
}

void emul_compress(const uint8_t* aff, uint8_t* out) {
  Aff a;
  memcpy(&a, aff, sizeof a);
  aff_compress(a, out);
}
void emul_to_affine(const uint8_t* jac, uint8_t* out) {
  Jac p;
  memcpy(&p, jac, sizeof p);
  Aff a = p.is_identity() ? Aff::identity() : jac_to_affine_with_zinv(p, fe_inv(p.z));
  memcpy(out, &a, sizeof a);
}

// engine transcript: merlin's published vector and a curdleproofs-style challenge
void emul_merlin_vector(uint8_t out[32]) {
  host::Transcript t("test protocol");
  t.append_message("some label", (const uint8_t*)"some data", 9);
  t.challenge_bytes("challenge", out, 32);
}
// hashes `len` bytes under "lbl", then returns two successive challenges (Montgomery wire form)
void emul_challenges(const uint8_t* msg, size_t len, uint8_t out[64]) {
  host::Transcript t("curdleproofs");
  t.append_message("lbl", msg, len);
  host::S a = t.get_and_append_challenge("ch"), b = t.get_and_append_challenge("ch");
  memcpy(out, a.f.v, 32);
  memcpy(out + 32, b.f.v, 32);
}
void emul_keccak(uint64_t* st) { host::keccak_f1600(st); }

}  // extern "C"
