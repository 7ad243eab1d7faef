// TEST-ONLY: compiles the product's host/device-portable math (curdleproofs_amd/csrc/*.hpp,
// host_math.hpp) for the CPU so that `pytest -m "not gpu"` can check the exact limb arithmetic,
// point formulas, scalar recodings and the engine's transcript against the oracle without a GPU.
// This is NOT a product path: nothing in curdleproofs_amd/ links or loads it.
#include <algorithm>
#include <cstring>
#include <vector>
#include "../../curdleproofs_amd/csrc/g1.hpp"
#include "../../curdleproofs_amd/csrc/g1_28.hpp"
#include "../../curdleproofs_amd/csrc/modinv30.hpp"
#include "../../curdleproofs_amd/csrc/glv.hpp"
#include "../../curdleproofs_amd/csrc/recode.hpp"
#include "../../curdleproofs_amd/csrc/host_math.hpp"
#include "../../curdleproofs_amd/csrc/quad_program.hpp"
#include "../../curdleproofs_amd/csrc/host_threads.hpp"
#include <stdexcept>

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
// the host engine's own Fr product (64-bit limbs, host_math.hpp)
void emul_host_fr_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    host::S x, y;
    memcpy(x.f.v, a + 32 * i, 32);
    memcpy(y.f.v, b + 32 * i, 32);
    const host::S z = x * y;
    memcpy(out + 32 * i, z.f.v, 32);
  }
}
// Euclid inverse of the host Fr against the Fermat chain; returns the number of mismatches
int emul_host_fr_inv_check(const uint8_t* a, size_t n) {
  int bad = 0;
  for (size_t i = 0; i < n; i++) {
    host::S x;
    memcpy(x.f.v, a + 32 * i, 32);
    const host::S e = x.inverse(), f = x.inverse_fermat();
    if (memcmp(e.f.v, f.f.v, 32) != 0) bad++;
    if (!x.is_zero() && !((e * x) == host::S::one())) bad++;
  }
  return bad;
}
void emul_fr_inv(const uint8_t* a, uint8_t* out) {
  host::S s;
  memcpy(s.f.v, a, 32);
  s = s.inverse();
  memcpy(out, s.f.v, 32);
}

// the MSM exactly as k_to_table + k_msm_accw + k_msm_tail(8,4) + k_msm_tail(8,32) compute it
void emul_msm(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t* out) {
  const Aff* P = reinterpret_cast<const Aff*>(bases);
  std::vector<TJac> bucket(512, TJac::identity());
  std::vector<int8_t> dig(64);
  for (size_t i = 0; i < n; i++) {
    Fr k;
    memcpy(k.v, scalars + 32 * i, 32);
    k = fe_from_mont(k);
    recode_signed16(k.v, dig.data());
    const TAff q = t_from_std(P[i]);                    // k_to_table
    for (int w = 0; w < 64; w++) {
      int d = dig[w];
      if (!d) continue;
      int mag = d < 0 ? -d : d;
      TJac& b = bucket[w * 8 + mag - 1];
      b = t_add_mixed(b, t_cneg(q, d < 0));
    }
  }
  TJac wsum[64];
  for (int w = 0; w < 64; w++) {
    TJac run = TJac::identity(), tot = TJac::identity();
    for (int b = 7; b >= 0; b--) {
      run = t_add(run, bucket[w * 8 + b]);
      tot = t_add(tot, run);
    }
    wsum[w] = tot;
  }
  auto tail = [](const TJac* in, TJac* o, int nout, int group, int shift) {
    for (int t = 0; t < nout; t++) {
      const TJac* src = in + t * group;
      TJac acc = src[group - 1];
      for (int j = group - 2; j >= 0; j--) {
        for (int s = 0; s < shift; s++) acc = t_dbl(acc);
        acc = t_add(acc, src[j]);
      }
      o[t] = acc;
    }
  };
  TJac part[8], res28;
  tail(wsum, part, 8, 8, 4);
  tail(part, &res28, 1, 8, 32);
  const Jac res = t_jac_to_std(res28);
  memcpy(out, &res, sizeof res);
}

// k_smul's per-element computation: (add ? add : O) + s*mul via the endomorphism split + NAF of both halves, Jacobian (standard form) out
void emul_smul(const uint8_t* add, const uint8_t* mul, const uint8_t* scalar, uint8_t* out) {
  Fr k;
  memcpy(k.v, scalar, 32);
  k = fe_from_mont(k);
  SmulNaf rn;
  recode_smul_glv(k.v, rn);
  Aff Ps;
  memcpy(&Ps, mul, sizeof Ps);
  const TAff P = t_from_std(Ps);
  const TAff NP = P.is_identity() ? P : TAff{t_mul(P.x, t_beta()), t_neg(P.y)};
  TJac acc = TJac::identity();
  for (int i = 128; i >= 0; i--) {
    acc = t_dbl(acc);
    if ((rn.nz[0][i >> 5] >> (i & 31)) & 1u) acc = t_add_mixed(acc, t_cneg(P, (rn.ng[0][i >> 5] >> (i & 31)) & 1u));
    if ((rn.nz[1][i >> 5] >> (i & 31)) & 1u) acc = t_add_mixed(acc, t_cneg(NP, (rn.ng[1][i >> 5] >> (i & 31)) & 1u));
  }
  if (add) {
    Aff A;
    memcpy(&A, add, sizeof A);
    acc = t_add_mixed(acc, t_from_std(A));
  }
  const Jac r = t_jac_to_std(acc);
  memcpy(out, &r, sizeof r);
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

// ---- 28-bit-limb lazy field (fp28.hpp / g1_28.hpp) ----
static double f28_mag(const F28& a) {   // |value| / p, approximately
  long double v = 0, pp = 0;
  for (int i = 13; i >= 0; i--) {
    v = v * 268435456.0L + a.v[i];
    pp = pp * 268435456.0L + F28Cfg::P[i];
  }
  return (double)(v < 0 ? -v / pp : v / pp);
}
void emul_f28_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Fp x, y;
    memcpy(x.v, a + 48 * i, 48);
    memcpy(y.v, b + 48 * i, 48);
    Fp z = f28_to_std(f28_mul(f28_from_std(x), f28_from_std(y)));
    memcpy(out + 48 * i, z.v, 48);
  }
}
// a b - c d with one reduction (f28_mulsub_body), operands blown up to `scale` x their canonical value by lazy additions first
void emul_f28_mulsub(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, uint8_t* out, size_t n, int scale) {
  for (size_t i = 0; i < n; i++) {
    Fp x[4];
    memcpy(x[0].v, a + 48 * i, 48);
    memcpy(x[1].v, b + 48 * i, 48);
    memcpy(x[2].v, c + 48 * i, 48);
    memcpy(x[3].v, d + 48 * i, 48);
    F28 v[4];
    for (int j = 0; j < 4; j++) {
      const F28 base = f28_from_std(x[j]);
      v[j] = base;
      for (int k = 1; k < scale; k++) v[j] = f28_add(v[j], base);   // scale * value, unreduced
    }
    Fp z = f28_to_std(f28_mulsub_body(v[0], v[1], v[2], v[3]));
    memcpy(out + 48 * i, z.v, 48);
  }
}
// lazy chain: ((x + y) * 8 - y * 2) * (x - y) - x * x, and the inverse of x
void emul_f28_lazy(const uint8_t* a, const uint8_t* b, uint8_t* out /*2*48*/) {
  Fp x, y;
  memcpy(x.v, a, 48);
  memcpy(y.v, b, 48);
  const F28 X = f28_from_std(x), Y = f28_from_std(y);
  const F28 t = f28_sub(f28_shl<3>(f28_add(X, Y)), f28_shl<1>(Y));
  const F28 r = f28_sub(f28_mul(t, f28_sub(X, Y)), f28_sqr(X));
  Fp o0 = f28_to_std(f28_mul(r, f28_one())), o1 = f28_to_std(f28_inv(f28_neg(X)));
  memcpy(out, o0.v, 48);
  memcpy(out + 48, o1.v, 48);
}
// sum k_i P_i by per-point double-and-add (mixed adds of table entries), then a Jacobian add chain; returns
// the affine result in standard form and the largest coordinate magnitude seen (units of p) in mags[3]
void emul_f28_msm(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t* out, double* mags) {
  const Aff* P = reinterpret_cast<const Aff*>(bases);
  Jac28 total = Jac28::identity();
  double mx = 0, my = 0, mz = 0;
  auto track = [&](const Jac28& j) {
    mx = std::max(mx, f28_mag(j.x)); my = std::max(my, f28_mag(j.y)); mz = std::max(mz, f28_mag(j.z));
  };
  for (size_t i = 0; i < n; i++) {
    Fr k;
    memcpy(k.v, scalars + 32 * i, 32);
    const Fr ks = fe_from_mont(k);
    const Aff28 q = aff28_from_std(P[i]);
    Jac28 acc = Jac28::identity();
    for (int b = 254; b >= 0; b--) {
      acc = jac28_dbl(acc);
      track(acc);
      if ((ks.v[b >> 5] >> (b & 31)) & 1u) {
        acc = jac28_add_mixed(acc, (b & 1) ? aff28_cneg(aff28_cneg(q, true), true) : q);
        track(acc);
      }
    }
    total = jac28_add(total, acc);
    track(total);
  }
  Aff r = Aff::identity();
  if (!total.is_identity()) r = aff28_to_std(jac28_to_affine_with_zinv(total, f28_inv(total.z)));
  memcpy(out, &r, 96);
  mags[0] = mx; mags[1] = my; mags[2] = mz;
}
// bucket accumulation in XYZZ coordinates: sum of +-P_i (sign from the low bit of the scalar's first limb), with
// repeated and opposite points mixed in; returns the affine sum (standard form) and the largest magnitudes seen
void emul_f28_xyzz_sum(const uint8_t* bases, size_t n, const uint8_t* signs, uint8_t* out, double* mags) {
  const Aff* P = reinterpret_cast<const Aff*>(bases);
  Xyzz28 acc = Xyzz28::identity();
  double mx = 0, my = 0;
  for (size_t i = 0; i < n; i++) {
    // alternate between the called-product form and the inlined form of the bucket loops (Y3 with one shared reduction)
    acc = (i & 1) ? xyzz28_add_mixed_t<true>(acc, aff28_cneg_lazy(aff28_from_std(P[i]), signs[i] & 1))   // as the bucket loops call it
                  : xyzz28_add_mixed(acc, aff28_cneg(aff28_from_std(P[i]), signs[i] & 1));
    mx = std::max(mx, f28_mag(acc.x));
    my = std::max(my, f28_mag(acc.y));
  }
  const Jac28 j = xyzz28_to_jac(acc);
  Aff r = Aff::identity();
  if (!j.is_identity()) r = aff28_to_std(jac28_to_affine_with_zinv(j, f28_inv(j.z)));
  memcpy(out, &r, 96);
  mags[0] = mx;
  mags[1] = my;
}
// the reduction kernels' full XYZZ addition (xyzz28_add, complete): the signed sum of n points, each first brought to a
// representation with ZZ != 1, added up front to back and as a tree; then the exceptional cases on point 0:
// out = [sum (chain), sum (tree), A + A, A + (-A), O + A, A + O, 2A (xyzz28_dbl), (A + A) + A]
void emul_f28_xyzz_full(const uint8_t* bases, size_t n, const uint8_t* signs, uint8_t* out /* 8 * 96 */, double* mags) {
  const Aff* P = reinterpret_cast<const Aff*>(bases);
  std::vector<Xyzz28> v(n);
  double mx = 0, my = 0;
  for (size_t i = 0; i < n; i++) {
    const Aff28 a = aff28_cneg(aff28_from_std(P[i]), signs[i] & 1);
    Xyzz28 t = xyzz28_add_mixed(xyzz28_add_mixed(Xyzz28::identity(), a), a);   // 2A
    v[i] = xyzz28_add_mixed(t, aff28_cneg(a, true));                              // A with ZZ != 1
  }
  Xyzz28 chain = Xyzz28::identity();
  for (size_t i = 0; i < n; i++) {
    chain = xyzz28_add(chain, v[i]);
    mx = std::max(mx, f28_mag(chain.x));
    my = std::max(my, f28_mag(chain.y));
  }
  std::vector<Xyzz28> tr = v;
  for (size_t m = n; m > 1; m = (m + 1) / 2)
    for (size_t i = 0; i < m / 2; i++) tr[i] = xyzz28_add(tr[i], tr[m - 1 - i]);
  const Xyzz28 A = v[0], nA = Xyzz28{A.x, f28_neg(A.y), A.zz, A.zzz};
  const Xyzz28 res[8] = {chain, n ? tr[0] : Xyzz28::identity(), xyzz28_add(A, A), xyzz28_add(A, nA), xyzz28_add(Xyzz28::identity(), A),
                         xyzz28_add(A, Xyzz28::identity()), xyzz28_dbl(A), xyzz28_add(xyzz28_add(A, A), A)};
  for (int k = 0; k < 8; k++) {
    const Jac28 j = xyzz28_to_jac(res[k]);
    Aff r = Aff::identity();
    if (!j.is_identity()) r = aff28_to_std(jac28_to_affine_with_zinv(j, f28_inv(j.z)));
    memcpy(out + 96 * k, &r, 96);
  }
  mags[0] = mx;
  mags[1] = my;
}
// exceptional cases of the complete formulas: P+P, P+(-P), P+O, O+P in mixed and Jacobian form -> 8 affine outputs
void emul_f28_edges(const uint8_t* base, uint8_t* out /*8*96*/) {
  Aff p;
  memcpy(&p, base, 96);
  const Aff28 q = aff28_from_std(p), nq = aff28_cneg(q, true);
  Jac28 j = jac28_dbl(jac28_dbl(Jac28::from_affine(q)));           // 4P with Z != 1
  const Jac28 j3 = jac28_add_mixed(j, nq);                           // 3P
  const Jac28 j4b = jac28_add_mixed(j3, q);                          // 4P via another route (different Z)
  Jac28 res[8] = {
      jac28_add_mixed(Jac28::from_affine(q), q),                     // 2P via mixed doubling branch
      jac28_add_mixed(Jac28::from_affine(q), nq),                    // O
      jac28_add_mixed(j, Aff28::identity()),                         // 4P
      jac28_add_mixed(Jac28::identity(), q),                         // P
      jac28_add(j, j4b),                                             // 8P via Jacobian doubling branch
      jac28_add(j, Jac28{j4b.x, f28_neg(j4b.y), j4b.z}),             // O
      jac28_add(j, Jac28::identity()),                               // 4P
      jac28_add(Jac28::identity(), j3)};                             // 3P
  for (int i = 0; i < 8; i++) {
    Aff r = Aff::identity();
    if (!res[i].is_identity()) r = aff28_to_std(jac28_to_affine_with_zinv(res[i], f28_inv(res[i].z)));
    memcpy(out + 96 * i, &r, 96);
  }
}

// decompression helpers on the table form: sqrt candidate and the endomorphism subgroup test
void emul_f28_sqrt(const uint8_t* a, uint8_t* out) {
  Fp x;
  memcpy(x.v, a, 48);
  Fp r = f28_to_std(f28_sqrt_candidate(f28_from_std(x)));
  memcpy(out, r.v, 48);
}
int emul_f28_in_subgroup(const uint8_t* aff) {
  Aff p;
  memcpy(&p, aff, 96);
  return g1_28_in_subgroup(aff28_from_std(p)) ? 1 : 0;
}

// binary-Euclid inversions against Fermat, both field forms
int emul_inv_euclid(const uint8_t* a, uint8_t* out /*2*48: std form, table form converted back*/) {
  Fp x;
  memcpy(x.v, a, 48);
  const Fp e = fe_inv_euclid(x), f = fe_inv(x);
  const F28 x28 = f28_mul(f28_from_std(x), f28_one());
  const Fp e28 = f28_to_std(f28_inv_euclid(x28)), f28v = f28_to_std(f28_inv(x28));
  memcpy(out, e.v, 48);
  memcpy(out + 48, e28.v, 48);
  return (memcmp(e.v, f.v, 48) == 0 && memcmp(e28.v, f28v.v, 48) == 0 && memcmp(e.v, e28.v, 48) == 0) ? 1 : 0;
}

// division-step inversion (modinv30.hpp) against the binary Euclidean one, canonical integers; returns mismatches
int emul_modinv30_check(const uint8_t* a, size_t n, uint8_t* first_out) {
  int bad = 0;
  for (size_t i = 0; i < n; i++) {
    uint32_t w[12], x[12], y[12];
    memcpy(w, a + 48 * i, 48);
    words_inv_mod_p_divsteps(w, x);
    words_inv_mod_p(w, y);
    if (i == 0) memcpy(first_out, x, 48);
    if (memcmp(x, y, 48) != 0) bad++;
  }
  return bad;
}

// the scalar-field instance (fr_inv_divsteps: Montgomery in, Montgomery out): out[i] = a[i]^-1 (wire form)
void emul_fr_inv_divsteps(const uint8_t* a, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; i++) {
    Fr x, y;
    memcpy(x.v, a + 32 * i, 32);
    y = fr_inv_divsteps(x);
    memcpy(out + 32 * i, y.v, 32);
  }
}

// endomorphism split of canonical scalars (glv.hpp): per scalar |t| (16 B), q (16 B), neg_k, neg_t (1 B each) and the
// 32 biased digit bytes the table kernel derives from them
void emul_glv_split(const uint8_t* k, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; i++) {
    uint32_t w[8], t[4], q[4], nk, nt, bt[4], bq[4];
    memcpy(w, k + 32 * i, 32);
    glv_split(w, t, q, nk, nt);
    glv_biased_bytes(t, bt);
    glv_biased_bytes(q, bq);
    uint8_t* o = out + 66 * i;
    memcpy(o, t, 16);
    memcpy(o + 16, q, 16);
    o[32] = (uint8_t)nk;
    o[33] = (uint8_t)nt;
    memcpy(o + 34, bt, 16);
    memcpy(o + 50, bq, 16);
  }
}

// k_table_build (endo rows) + k_msm_tblw + k_reduce_sets / k_reduce_groups on one task, in the kernels' own arithmetic:
// 16 shifted copies 2^(8c) P and their images (beta x, -y), scalars split by glv_split, 128 XYZZ buckets shared by all 32
// windows, then per half the running sums over groups of 8 and over the 8 groups.  perwin != 0: the verifier's variant
// (k_to_table_endo + k_msm_tblw<2, true>): no shifted copies, one bucket set pair per window, Horner over the 16 windows.
void emul_msm_endo(const uint8_t* bases, const uint8_t* scalars, size_t n, int perwin, uint8_t* out) {
  const Aff* P = reinterpret_cast<const Aff*>(bases);
  const TF beta = t_beta();
  const int nsetpairs = perwin ? 16 : 1;
  std::vector<TAcc> bucket((size_t)nsetpairs * 128, TAcc::identity());
  for (size_t i = 0; i < n; i++) {
    // table row of point i
    TAff row[32];
    const TAff p0 = t_from_std(P[i]);
    const bool inf = p0.is_identity();
    TJac acc = TJac::from_affine(p0);
    for (int c = 0; c < 16; c++) {
      TAff a = p0;
      if (c && !inf) {
        for (int d = 0; d < 8; d++) acc = t_dbl(acc);
        a = t_to_affine(acc, t_inv(acc.z));
      }
      row[c] = a;
      row[16 + c] = inf ? a : TAff{t_mul(a.x, beta), t_neg(a.y)};
    }
    Fr k;
    memcpy(k.v, scalars + 32 * i, 32);
    k = fe_from_mont(k);
    uint32_t t[4], q[4], bytes[8], nk, nt;
    glv_split(k.v, t, q, nk, nt);
    glv_biased_bytes(t, bytes);
    glv_biased_bytes(q, bytes + 4);
    for (int w = 0; w < 32; w++) {
      const int d = (int)((bytes[w >> 2] >> (8 * (w & 3))) & 255u) - 128;
      if (!d) continue;
      const bool neg = (d < 0) != ((w < 16 ? (nk ^ nt) : nk) != 0);
      const int b = (d < 0 ? -d : d) - 1;
      // perwin: copy 0 / 1 of the unshifted point, bucket set pair = window; else copy w, one shared set pair
      const TAff src = perwin ? row[w < 16 ? 0 : 16] : row[w];
      TAcc& B = bucket[(size_t)(perwin ? (w & 15) : 0) * 128 + b];
      B = t_acc_add_mixed(B, t_cneg(src, neg));
    }
  }
  auto reduce_set = [&](const TAcc* A, int upper) {   // k_reduce_sets<true> + k_reduce_groups<true>
    TJac S[8], W[8];
    for (int g = 0; g < 8; g++) {
      TJac run = t_acc_to_jac(A[g * 8 + 7]), w = run;
      for (int j = 6; j >= 0; j--) {
        run = t_add(run, t_acc_to_jac(A[g * 8 + j]));
        w = t_add(w, run);
      }
      S[g] = run;
      W[g] = w;
    }
    TJac run = S[7], v = run;
    for (int g = 6; g >= 1; g--) {
      run = t_add(run, S[g]);
      v = t_add(v, run);
    }
    run = t_add(run, S[0]);
    for (int d = 0; d < 3; d++) v = t_dbl(v);
    for (int g = 0; g < 8; g++) v = t_add(v, W[g]);
    if (upper) {
      for (int d = 0; d < 6; d++) run = t_dbl(run);
      v = t_add(v, run);
    }
    return v;
  };
  TJac res = TJac::identity();
  for (int w = nsetpairs - 1; w >= 0; w--) {   // Horner over the windows (one iteration for the table form)
    if (perwin)
      for (int d = 0; d < 8; d++) res = t_dbl(res);
    res = t_add(res, reduce_set(&bucket[(size_t)w * 128], 0));
    res = t_add(res, reduce_set(&bucket[(size_t)w * 128 + 64], 1));
  }
  const Jac r = t_jac_to_std(res);
  memcpy(out, &r, sizeof r);
}

// ---- late rounds (late.hip), in the kernels' own arithmetic and with the product's recodings (recode.hpp) ----
// Fixed-base digit recodings: for every scalar the W = ceil(256 / cb) digits of FixDigitStream<cb> (k_late_fix) and of
// fix_window_digits<cb> (k_msm_fix) as int32, stream first: out[i][2][32]
}  // extern "C" (templates below)
template <int CB> static void fix_digits_both(const uint32_t* k, int32_t* o) {
  FixDigitStream<CB> ds(k);
  for (int w = 0; w < FixWin<CB>::W; w++) o[w] = ds.next();
  if (CB <= 16) {
    int16_t d[32];
    fix_window_digits<CB, FixWin<CB>::W>(k, 0, d, 1);
    for (int w = 0; w < FixWin<CB>::W; w++) o[32 + w] = d[w];
  } else {
    fix_window_digits<CB, FixWin<CB>::W>(k, 0, o + 32, 1);
  }
}
extern "C" {
int emul_fix_digits(int cb, const uint8_t* k, size_t n, int32_t* out) {
  for (size_t i = 0; i < n; i++) {
    uint32_t w[8];
    memcpy(w, k + 32 * i, 32);
    int32_t* o = out + 64 * i;
    memset(o, 0, 64 * sizeof(int32_t));
    if (cb == 8) fix_digits_both<8>(w, o);
    else if (cb == 16) fix_digits_both<16>(w, o);
    else if (cb == 19) fix_digits_both<19>(w, o);
    else return -1;
  }
  return cb == 8 ? FixWin<8>::W : cb == 16 ? FixWin<16>::W : FixWin<19>::W;
}
// signed radix-16 digits of values < 2^127 (16 bytes each): out[i][32] = digit in [-7, 8]
void emul_signed_nibbles(const uint8_t* v, size_t n, int8_t* out) {
  for (size_t i = 0; i < n; i++) {
    uint32_t w[4], packed[4];
    memcpy(w, v + 16 * i, 16);
    recode_signed_nibbles_biased(w, packed);
    for (int j = 0; j < 32; j++) out[32 * i + j] = (int8_t)((int)((packed[j >> 3] >> (4 * (j & 7))) & 15u) - 7);
  }
}
namespace {
// table row of a point as k_table_build (endo) leaves it: 16 shifted copies 2^(8c) P and their images (beta x, -y)
void emul_shifted_row(const Aff& P, TAff* row /*32*/) {
  const TF beta = t_beta();
  const TAff p0 = t_from_std(P);
  const bool inf = p0.is_identity();
  TJac acc = TJac::from_affine(p0);
  for (int c = 0; c < 16; c++) {
    TAff a = p0;
    if (c && !inf) {
      for (int d = 0; d < 8; d++) acc = t_dbl(acc);
      a = t_to_affine(acc, t_inv(acc.z));
    }
    row[c] = a;
    row[16 + c] = inf ? a : TAff{t_mul(a.x, beta), t_neg(a.y)};
  }
}
}  // namespace
// k_late_uniform for ONE output: sum_t S_t * P_t with the shared digit sequence — entries (t, copy) counting-sorted by |digit|
// (descending), `split` lanes taking every split-th entry of a level, each with its own running sum and total, totals added.
void emul_late_uniform(const uint8_t* bases, const uint8_t* scalars, size_t nt, int split, uint8_t* out) {
  const Aff* P = reinterpret_cast<const Aff*>(bases);
  std::vector<TAff> rows(32 * nt);
  for (size_t t = 0; t < nt; t++) emul_shifted_row(P[t], &rows[32 * t]);
  std::vector<std::vector<uint32_t>> level(129);   // entries per magnitude: (t << 6) | (copy << 1) | neg
  for (size_t t = 0; t < nt; t++) {
    Fr k;
    memcpy(k.v, scalars + 32 * t, 32);
    k = fe_from_mont(k);
    uint32_t tt[4], qq[4], bytes[8], nk, ntg;
    glv_split(k.v, tt, qq, nk, ntg);
    glv_biased_bytes(tt, bytes);
    glv_biased_bytes(qq, bytes + 4);
    for (int c = 0; c < 32; c++) {
      const int d = (int)((bytes[c >> 2] >> (8 * (c & 3))) & 255u) - 128;
      if (!d) continue;
      const uint32_t neg = (d < 0 ? 1u : 0u) ^ (c < 16 ? (nk ^ ntg) : nk);
      level[d < 0 ? -d : d].push_back(((uint32_t)t << 6) | ((uint32_t)c << 1) | neg);
    }
  }
  std::vector<TAcc> acc(split, TAcc::identity()), total(split, TAcc::identity());
  size_t pos = 0;
  for (int b = 128; b >= 1; b--) {
    const size_t c = level[b].size();
    for (int sub = 0; sub < split; sub++) {
      for (size_t e = (size_t)sub; e < c; e += (size_t)split) {
        const uint32_t en = level[b][e];
        acc[sub] = t_acc_add_mixed(acc[sub], t_cneg(rows[32 * (en >> 6) + ((en >> 1) & 31u)], (en & 1u) != 0));
      }
      if (pos + c) total[sub] = t_acc_add(total[sub], acc[sub]);
    }
    pos += c;
  }
  TAcc sum = total[0];
  for (int sub = 1; sub < split; sub++) sum = t_acc_add(sum, total[sub]);
  const Jac r = t_jac_to_std(t_acc_to_jac(sum));
  memcpy(out, &r, sizeof r);
}
// k_late_tables + k_late_msm + k_msm_tail for ONE output: multiples 1..8 of every point and their images, Straus over the
// endomorphism-split scalars with signed 4-bit digits, the 32 windows in `slices` groups joined by a Horner chain.
void emul_late_msm(const uint8_t* bases, const uint8_t* scalars, size_t n, int slices, uint8_t* out) {
  const Aff* P = reinterpret_cast<const Aff*>(bases);
  const TF beta = t_beta();
  std::vector<TAff> tab(16 * n);
  for (size_t i = 0; i < n; i++) {
    const TJac P0 = TJac::from_affine(t_from_std(P[i]));
    TJac m[8];
    for (int j = 0; j < 8; j++) m[j] = j == 0 ? P0 : (j & 1) ? t_dbl(m[j / 2]) : t_add(m[j - 1], P0);
    for (int j = 0; j < 8; j++) {
      const TAff a = m[j].is_identity() ? TAff::identity() : t_to_affine(m[j], t_inv(m[j].z));
      tab[16 * i + j] = a;
      tab[16 * i + 8 + j] = a.is_identity() ? a : TAff{t_mul(a.x, beta), t_neg(a.y)};
    }
  }
  std::vector<uint32_t> dig(8 * n);
  std::vector<uint32_t> sign(2 * n);
  for (size_t i = 0; i < n; i++) {
    Fr k;
    memcpy(k.v, scalars + 32 * i, 32);
    k = fe_from_mont(k);
    uint32_t h[2][4], nk, ntg;
    glv_split(k.v, h[0], h[1], nk, ntg);
    sign[2 * i] = nk ^ ntg;
    sign[2 * i + 1] = nk;
    recode_signed_nibbles_biased(h[0], &dig[8 * i]);
    recode_signed_nibbles_biased(h[1], &dig[8 * i + 4]);
  }
  const int ws = 32 / slices;
  std::vector<TJac> part(slices);
  for (int sl = 0; sl < slices; sl++) {
    TAcc acc = TAcc::identity();
    for (int win = sl * ws + ws - 1; win >= sl * ws; win--) {
      if (win != sl * ws + ws - 1)
        for (int q = 0; q < 4; q++) acc = t_acc_dbl(acc);
      for (size_t sh = 0; sh < 2 * n; sh++) {
        const int d = (int)((dig[4 * sh + (win >> 3)] >> (4 * (win & 7))) & 15u) - 7;
        if (!d) continue;
        const bool neg = (d < 0) != (sign[sh] != 0);
        acc = t_acc_add_mixed(acc, t_cneg(tab[16 * (sh >> 1) + 8 * (sh & 1) + (d < 0 ? -d : d) - 1], neg));
      }
    }
    part[sl] = t_acc_to_jac(acc);
  }
  TJac res = part[slices - 1];   // k_msm_tail: group = slices, shift = 128 / slices
  for (int j = slices - 2; j >= 0; j--) {
    for (int d = 0; d < 128 / slices; d++) res = t_dbl(res);
    res = t_add(res, part[j]);
  }
  const Jac r = t_jac_to_std(res);
  memcpy(out, &r, sizeof r);
}

// The reduction programs of k_reduce_sets_wave (quad_program.hpp: the tables the GPU runs) interpreted over integers: a point is
// a number, an addition a sum, a doubling a product by two.  kind 0 = plain, 1 = bucket set (lower half), 2 = upper half.
// out[0] = the result, out[1] = rounds, out[2] = addition rounds, out[3] = hazards (a job reading an entry another job of the
// same round writes, two jobs writing one entry, an index outside the array).
void emul_quad_program(int kind, const int64_t* in /* 64 */, int64_t* out /* 4 */) {
  static constexpr QuadProgram progs[3] = {qp_plain(), qp_weighted(false), qp_weighted(true)};
  const QuadProgram& p = progs[kind];
  int64_t buf[QP_ENTRIES];
  for (int i = 0; i < QP_ENTRIES; i++) buf[i] = i < 64 ? in[i] : 0x5555555555ll;   // scratch entries start as garbage
  buf[QP_IDENT] = 0;
  int64_t hazards = 0, add_rounds = 0;
  for (int r = 0; r < p.nrounds; r++) {
    int64_t res[16];
    bool written[QP_ENTRIES] = {};
    for (int q = 0; q < 16; q++) {
      const QuadJob j = p.job[r][q];
      if (j.a >= QP_ENTRIES || j.b >= QP_ENTRIES || j.dst >= QP_ENTRIES) hazards++;
      res[q] = p.dbl[r] ? 2 * buf[j.a] : buf[j.a] + buf[j.b];
      if (j.on) {
        if (written[j.dst]) hazards++;
        written[j.dst] = true;
      }
    }
    for (int q = 0; q < 16; q++) {   // a read of an entry that ANOTHER job of the round writes depends on the quads' order
      const QuadJob j = p.job[r][q];
      if (written[j.a] && !(j.on && j.dst == j.a)) hazards++;
      if (!p.dbl[r] && written[j.b] && !(j.on && j.dst == j.b)) hazards++;
    }
    for (int q = 0; q < 16; q++)
      if (p.job[r][q].on) buf[p.job[r][q].dst] = res[q];
    add_rounds += p.dbl[r] ? 0 : 1;
  }
  out[0] = buf[p.result];
  out[1] = p.nrounds;
  out[2] = add_rounds;
  out[3] = hazards;
}

// The engine's thread helpers (host_threads.hpp).  kind 0 = WorkerPool, 1 = SpinTeam engaged, 2 = SpinTeam engaged / released
// between jobs.  `jobs` parallel loops over n items each; every item adds its index + job number to a per-item cell and must be
// visited exactly once; job number `throw_at` (if >= 0) throws from item 0 and the exception must reach the caller while the
// other jobs stay intact.  Returns 0 if everything added up, a positive code otherwise.
int emul_thread_helpers(int kind, size_t threads, size_t n, size_t jobs, long throw_at) {
  std::vector<std::atomic<uint64_t>> cell(n);
  for (auto& c : cell) c.store(0);
  uint64_t expect = 0;
  int caught = 0;
  auto body = [&](size_t job) {
    const std::function<void(size_t)> f = [&, job](size_t i) {
      if ((long)job == throw_at && i == 0) throw std::runtime_error("item failed");
      cell[i].fetch_add(i + job + 1);
    };
    return f;
  };
  if (kind == 0) {
    WorkerPool pool(threads);
    for (size_t j = 0; j < jobs; j++) {
      try {
        pool.run(n, body(j));
      } catch (const std::runtime_error&) {
        caught++;
      }
    }
  } else {
    SpinTeam team(threads - 1);
    if (kind == 1) team.engage();
    for (size_t j = 0; j < jobs; j++) {
      if (kind == 2) team.engage();
      try {
        team.run(n, body(j));
      } catch (const std::runtime_error&) {
        caught++;
      }
      if (kind == 2) team.release();
    }
    team.release();
  }
  if (caught != (throw_at >= 0 && (size_t)throw_at < jobs ? 1 : 0)) return 1;
  for (size_t i = 0; i < n; i++) {
    expect = 0;
    for (size_t j = 0; j < jobs; j++)
      if (!((long)j == throw_at && i == 0)) expect += i + j + 1;
    // (a WorkerPool job that throws stops handing out further items: only check the failed job's cells loosely)
    const uint64_t got = cell[i].load();
    if (throw_at < 0 ? got != expect : (got > expect || got + (i + (uint64_t)throw_at + 1) < expect)) return 2;
  }
  return 0;
}

}  // extern "C"
