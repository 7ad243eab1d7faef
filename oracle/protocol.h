// ORACLE (test infrastructure — never linked into the product path).
//
// Straight CPU restatement of the Curdleproofs protocol code of the reference:
//   crs.rs:37-75               CurdleproofsCrs::{from_points, generate_crs, log2_n}
//   util.rs:32-106             blinders, verification-scalar bit table, inner_product, permutation,
//                              shuffle_permute_and_commit_input
//   commitments.rs:44-79       GroupCommitment
//   msm_accumulator.rs:22-68   MsmAccumulator
//   inner_product_argument.rs  InnerProductProof::{new, verification_scalars, verify, serialize}
//   grand_product_argument.rs  GrandProductProof::{new, verify}
//   same_permutation_argument.rs, same_scalar_argument.rs, same_multiscalar_argument.rs
//   curdleproofs.rs:59-323     CurdleproofsProof::{new, verify, serialize, deserialize}
//   whisk.rs:44-65,106-263     trackers, tracker proofs, whisk shuffle proof wrappers
// Each function names the reference lines it follows. Algorithm choices (per-element
// double-and-add folds with per-element inversion, Pippenger MSM) are the reference's, so the
// timings of this file are the "port" CPU baseline of bench.py.
#pragma once
#include "field.h"
#include "g1.h"
#include "merlin.h"
#include "stdrng.h"
#include <functional>
#include <map>
#include <array>
#include <vector>

namespace orc {

static const size_t N_BLINDERS = 4;        // lib.rs:35
static const size_t CRS_EXTRA_POINTS = 3;  // crs.rs:15

typedef std::function<Fr()> FrDraw;   // stands for `Fr::rand(rng)` at the reference's call sites

// ---------------------------------------------------------------- util.rs
static inline std::vector<Fr> generate_blinders(const FrDraw& draw, size_t n) {   // util.rs:32-34
  std::vector<Fr> v(n);
  for (auto& x : v) x = draw();
  return v;
}
static inline Fr inner_product(const Fr* a, const Fr* b, size_t n) {   // util.rs:66-73
  Fr c = Fr::zero();
  for (size_t i = 0; i < n; i++) c += a[i] * b[i];
  return c;
}
static inline Fr inner_product(const std::vector<Fr>& a, const std::vector<Fr>& b) {
  return inner_product(a.data(), b.data(), a.size());
}
template <class T> static inline std::vector<T> get_permutation(const std::vector<T>& v, const std::vector<uint32_t>& perm) {   // util.rs:76-78
  std::vector<T> o(perm.size());
  for (size_t i = 0; i < perm.size(); i++) o[i] = v[perm[i]];
  return o;
}
// util.rs:40-64: bitstring[i] lists, in increasing j, the round indices j with bit (logn-1-j) of i set
static inline std::vector<std::vector<size_t>> get_verification_scalars_bitstring(size_t n, size_t logn) {
  std::vector<std::vector<size_t>> bs(n);
  for (size_t j = 0; j < logn; j++)
    for (size_t i = 0; i < n; i++)
      if ((i >> (logn - j - 1)) & 1) bs[i].push_back(j);
  return bs;
}
static inline void batch_inversion(std::vector<Fr>& v) {   // ark_ff::batch_inversion (zeros stay zero)
  for (auto& x : v)
    if (!x.is_zero()) x = x.inverse();
}

// ---------------------------------------------------------------- crs.rs
struct Crs {
  std::vector<G1Aff> vec_G, vec_H;
  G1 H, G_t, G_u;
  G1Aff G_sum, H_sum;
  size_t log2_n() const {   // crs.rs:71-74
    size_t n = vec_G.size() + vec_H.size(), l = 0;
    while ((size_t(1) << l) < n) l++;
    return l;
  }
};
static inline G1Aff sum_affine_points(const std::vector<G1Aff>& v) {   // util.rs:108-114
  G1 acc = G1::identity();
  for (auto& p : v) acc = g1_add_mixed(acc, p);
  return g1_to_affine(acc);
}
static inline bool crs_from_points(size_t ell, const std::vector<G1Aff>& points, Crs* crs) {   // crs.rs:37-58
  size_t n = ell + N_BLINDERS;
  if (points.size() < n + CRS_EXTRA_POINTS) return false;
  crs->vec_G.assign(points.begin(), points.begin() + ell);
  crs->vec_H.assign(points.begin() + ell, points.begin() + n);
  crs->G_sum = sum_affine_points(crs->vec_G);
  crs->H_sum = sum_affine_points(crs->vec_H);
  crs->H = G1::from_affine(points[n]);
  crs->G_t = G1::from_affine(points[n + 1]);
  crs->G_u = G1::from_affine(points[n + 2]);
  return true;
}
static inline Crs generate_crs(size_t ell) {   // crs.rs:61-69
  StdRng rng(0);
  std::vector<G1Aff> pts(ell + N_BLINDERS + CRS_EXTRA_POINTS);
  for (auto& p : pts) p = g1_to_affine(rand_g1(rng));
  Crs crs;
  crs_from_points(ell, pts, &crs);
  return crs;
}

// ---------------------------------------------------------------- commitments.rs
struct GroupCommitment {
  G1 T_1, T_2;
  static GroupCommitment make(const G1& crs_G, const G1& crs_H, const G1& T, const Fr& r) {   // commitments.rs:44-55
    return GroupCommitment{g1_mul(crs_G, r), g1_add(T, g1_mul(crs_H, r))};
  }
  GroupCommitment add(const GroupCommitment& o) const { return GroupCommitment{g1_add(T_1, o.T_1), g1_add(T_2, o.T_2)}; }   // :59-68
  GroupCommitment mul(const Fr& k) const { return GroupCommitment{g1_mul(T_1, k), g1_mul(T_2, k)}; }                          // :71-79
  bool eq(const GroupCommitment& o) const { return g1_eq(T_1, o.T_1) && g1_eq(T_2, o.T_2); }
};

// ---------------------------------------------------------------- msm_accumulator.rs
struct MsmAccumulator {
  typedef std::array<uint64_t, 13> Key;   // x limbs, y limbs, infinity
  G1 A_c = G1::identity();
  std::map<Key, std::pair<G1Aff, Fr>> base_scalar_map;

  static Key key_of(const G1Aff& p) {
    Key k{};
    if (p.inf) {
      k[12] = 1;
      return k;
    }
    memcpy(&k[0], p.x.v, 48);
    memcpy(&k[6], p.y.v, 48);
    return k;
  }
  void accumulate_check(const G1& C, const std::vector<Fr>& vec_x, const std::vector<G1Aff>& vec_V, const FrDraw& draw) {   // :38-52
    Fr random_factor = draw();
    A_c = g1_add(A_c, g1_mul(C, random_factor));
    size_t n = std::min(vec_x.size(), vec_V.size());   // Rust `zip`
    for (size_t i = 0; i < n; i++) {
      auto it = base_scalar_map.emplace(key_of(vec_V[i]), std::make_pair(vec_V[i], Fr::zero())).first;
      it->second.second += random_factor * vec_x[i];
    }
  }
  bool verify() const {   // :55-68
    std::vector<G1Aff> bases;
    std::vector<Fr> scalars;
    for (auto& kv : base_scalar_map) {
      bases.push_back(kv.second.first);
      scalars.push_back(kv.second.second);
    }
    return g1_sub(g1_msm(bases, scalars), A_c).is_identity();
  }
};

// ---------------------------------------------------------------- inner_product_argument.rs
struct InnerProductProof {
  G1 B_c, B_d;
  std::vector<G1> vec_L_C, vec_R_C, vec_L_D, vec_R_D;
  Fr c_final, d_final;
};

// :42-82
static inline void generate_ipa_blinders(const FrDraw& draw, const std::vector<Fr>& c, const std::vector<Fr>& d,
                                         std::vector<Fr>* out_r, std::vector<Fr>* out_z) {
  size_t n = c.size();
  std::vector<Fr> r = generate_blinders(draw, n);
  std::vector<Fr> z = generate_blinders(draw, n - 2);
  Fr omega = inner_product(r, d) + inner_product(z.data(), c.data(), n - 2);
  Fr delta = inner_product(r.data(), z.data(), n - 2);
  Fr inv_c = c[n - 2].inverse();
  Fr last_z = (r[n - 2] * inv_c * omega - delta) * ((-r[n - 2]) * inv_c * c[n - 1] + r[n - 1]).inverse();
  Fr penultimate_z = (-inv_c) * (last_z * c[n - 1] + omega);
  z.push_back(penultimate_z);
  z.push_back(last_z);
  *out_r = r;
  *out_z = z;
}

// :98-199
static inline InnerProductProof ipa_new(std::vector<G1Aff> crs_G_vec, std::vector<G1Aff> crs_G_prime_vec, const G1& crs_H,
                                        const G1& C, const G1& D, const Fr& z, std::vector<Fr> vec_c, std::vector<Fr> vec_d,
                                        Transcript& transcript, const FrDraw& draw) {
  size_t n = vec_c.size();
  InnerProductProof pf;
  std::vector<Fr> vec_r_c, vec_r_d;
  generate_ipa_blinders(draw, vec_c, vec_d, &vec_r_c, &vec_r_d);

  pf.B_c = g1_msm(crs_G_vec, vec_r_c);
  pf.B_d = g1_msm(crs_G_prime_vec, vec_r_d);

  transcript.append_g1("ipa_step1", C);
  transcript.append_g1("ipa_step1", D);
  transcript.append_fr("ipa_step1", z);
  transcript.append_g1("ipa_step1", pf.B_c);
  transcript.append_g1("ipa_step1", pf.B_d);
  Fr alpha = transcript.get_and_append_challenge("ipa_alpha");
  Fr beta = transcript.get_and_append_challenge("ipa_beta");

  for (size_t i = 0; i < n; i++) {
    vec_c[i] = vec_r_c[i] + alpha * vec_c[i];
    vec_d[i] = vec_r_d[i] + alpha * vec_d[i];
  }
  G1 H = g1_mul(crs_H, beta);

  while (n > 1) {
    n /= 2;
    Fr* c_L = vec_c.data();
    Fr* c_R = c_L + n;
    Fr* d_L = vec_d.data();
    Fr* d_R = d_L + n;
    G1Aff* G_L = crs_G_vec.data();
    G1Aff* G_R = G_L + n;
    G1Aff* Gp_L = crs_G_prime_vec.data();
    G1Aff* Gp_R = Gp_L + n;

    G1 L_C = g1_add(g1_msm(G_R, c_L, n), g1_mul(H, inner_product(c_L, d_R, n)));
    G1 L_D = g1_msm(Gp_L, d_R, n);
    G1 R_C = g1_add(g1_msm(G_L, c_R, n), g1_mul(H, inner_product(c_R, d_L, n)));
    G1 R_D = g1_msm(Gp_R, d_L, n);

    pf.vec_L_C.push_back(L_C);
    pf.vec_L_D.push_back(L_D);
    pf.vec_R_C.push_back(R_C);
    pf.vec_R_D.push_back(R_D);

    transcript.append_g1("ipa_loop", L_C);
    transcript.append_g1("ipa_loop", L_D);
    transcript.append_g1("ipa_loop", R_C);
    transcript.append_g1("ipa_loop", R_D);
    Fr gamma = transcript.get_and_append_challenge("ipa_gamma");
    Fr gamma_inv = gamma.inverse();

    for (size_t i = 0; i < n; i++) {
      c_L[i] += gamma_inv * c_R[i];
      d_L[i] += gamma * d_R[i];
      G_L[i] = g1_to_affine(g1_add(G1::from_affine(G_L[i]), g1_mul(G_R[i], gamma)));
      Gp_L[i] = g1_to_affine(g1_add(G1::from_affine(Gp_L[i]), g1_mul(Gp_R[i], gamma_inv)));
    }
  }
  pf.c_final = vec_c[0];
  pf.d_final = vec_d[0];
  return pf;
}

// :203-250
static inline bool ipa_verification_scalars(const InnerProductProof& pf, size_t n, Transcript& transcript, std::vector<Fr>* gam,
                                            std::vector<Fr>* gam_inv, std::vector<Fr>* vec_s, std::vector<Fr>* vec_inv_s) {
  size_t lg_n = pf.vec_L_C.size();
  if (lg_n >= 32) return false;
  if (n != (size_t(1) << lg_n)) return false;
  auto bits = get_verification_scalars_bitstring(n, lg_n);
  std::vector<Fr> challenges;
  for (size_t i = 0; i < lg_n; i++) {
    transcript.append_g1("ipa_loop", pf.vec_L_C[i]);
    transcript.append_g1("ipa_loop", pf.vec_L_D[i]);
    transcript.append_g1("ipa_loop", pf.vec_R_C[i]);
    transcript.append_g1("ipa_loop", pf.vec_R_D[i]);
    challenges.push_back(transcript.get_and_append_challenge("ipa_gamma"));
  }
  std::vector<Fr> challenges_inv = challenges;
  batch_inversion(challenges_inv);
  std::vector<Fr> s(n, Fr::one());
  for (size_t i = 0; i < n; i++)
    for (size_t j : bits[i]) s[i] *= challenges[j];
  std::vector<Fr> inv_s = s;
  batch_inversion(inv_s);
  *gam = challenges;
  *gam_inv = challenges_inv;
  *vec_s = s;
  *vec_inv_s = inv_s;
  return true;
}

// :264-326
static inline bool ipa_verify(const InnerProductProof& pf, const std::vector<G1Aff>& crs_G_vec, const G1& crs_H, const G1& C,
                              const G1& D, const Fr& z, const std::vector<Fr>& vec_u, Transcript& transcript,
                              MsmAccumulator& acc, const FrDraw& draw) {
  size_t n = crs_G_vec.size();
  transcript.append_g1("ipa_step1", C);
  transcript.append_g1("ipa_step1", D);
  transcript.append_fr("ipa_step1", z);
  transcript.append_g1("ipa_step1", pf.B_c);
  transcript.append_g1("ipa_step1", pf.B_d);
  Fr alpha = transcript.get_and_append_challenge("ipa_alpha");
  Fr beta = transcript.get_and_append_challenge("ipa_beta");

  std::vector<Fr> vec_gamma, vec_gamma_inv, vec_s, vec_inv_s;
  if (!ipa_verification_scalars(pf, n, transcript, &vec_gamma, &vec_gamma_inv, &vec_s, &vec_inv_s)) return false;
  if (pf.vec_R_C.size() != vec_gamma.size() || pf.vec_L_D.size() != vec_gamma.size() || pf.vec_R_D.size() != vec_gamma.size())
    return false;   // (Rust would panic in msm's assert_eq; a byte-deserialised proof can never get here)

  std::vector<Fr> vec_rhs_scalars(n);
  for (size_t i = 0; i < n; i++) vec_rhs_scalars[i] = pf.c_final * vec_s[i];
  vec_rhs_scalars.push_back(pf.c_final * pf.d_final * beta);
  std::vector<G1Aff> vec_G_H = crs_G_vec;
  vec_G_H.push_back(g1_to_affine(crs_H));

  G1 H = g1_mul(crs_H, beta);
  G1 C_a = g1_add(g1_add(pf.B_c, g1_mul(C, alpha)), g1_mul(H, alpha * alpha * z));
  G1 point_lhs = g1_add(g1_add(g1_msm_from_projective(pf.vec_L_C, vec_gamma), C_a), g1_msm_from_projective(pf.vec_R_C, vec_gamma_inv));
  acc.accumulate_check(point_lhs, vec_rhs_scalars, vec_G_H, draw);

  std::vector<Fr> vec_d_div_s(n);
  for (size_t i = 0; i < n; i++) vec_d_div_s[i] = pf.d_final * (vec_inv_s[i] * vec_u[i]);
  G1 D_a = g1_add(pf.B_d, g1_mul(D, alpha));
  point_lhs = g1_add(g1_add(g1_msm_from_projective(pf.vec_L_D, vec_gamma), D_a), g1_msm_from_projective(pf.vec_R_D, vec_gamma_inv));
  acc.accumulate_check(point_lhs, vec_d_div_s, crs_G_vec, draw);
  return true;
}

// ---------------------------------------------------------------- grand_product_argument.rs
struct GrandProductProof {
  G1 C;
  Fr r_p;
  InnerProductProof ipa_proof;
};

// :43-169
static inline GrandProductProof gprod_new(const std::vector<G1Aff>& crs_G_vec, const std::vector<G1Aff>& crs_H_vec, const G1& crs_U,
                                          const G1& B, const Fr& gprod_result, std::vector<Fr> vec_b,
                                          const std::vector<Fr>& vec_b_blinders, Transcript& transcript, const FrDraw& draw) {
  size_t n_blinders = vec_b_blinders.size(), ell = crs_G_vec.size(), n = ell + n_blinders;
  uint64_t ell_plus_one = ell + 1;

  transcript.append_g1("gprod_step1", B);
  transcript.append_fr("gprod_step1", gprod_result);
  Fr alpha = transcript.get_and_append_challenge("gprod_alpha");

  std::vector<Fr> vec_c;
  vec_c.push_back(Fr::one());
  for (size_t i = 0; i + 1 < ell; i++) vec_c.push_back(vec_c[i] * vec_b[i]);

  std::vector<Fr> vec_c_blinders = generate_blinders(draw, n_blinders);
  G1 C = g1_add(g1_msm(crs_G_vec, vec_c), g1_msm(crs_H_vec, vec_c_blinders));

  std::vector<Fr> vec_r_b_plus_alpha(n_blinders);
  for (size_t i = 0; i < n_blinders; i++) vec_r_b_plus_alpha[i] = vec_b_blinders[i] + alpha;
  Fr r_p = inner_product(vec_r_b_plus_alpha, vec_c_blinders);

  transcript.append_g1("gprod_step2", C);
  transcript.append_fr("gprod_step2", r_p);
  Fr beta = transcript.get_and_append_challenge("gprod_beta");
  Fr beta_inv = beta.inverse();

  // :88-102 rescaled bases G'_i = beta^{-(i+1)} G_i, H'_i = beta^{-(ell+1)} H_i
  std::vector<G1Aff> vec_G_prime;
  Fr pow_beta_inv = beta_inv;
  for (auto& G_i : crs_G_vec) {
    vec_G_prime.push_back(g1_to_affine(g1_mul(G_i, pow_beta_inv)));
    pow_beta_inv *= beta_inv;
  }
  std::vector<G1Aff> vec_H_prime;
  Fr beta_inv_pow = beta_inv.pow_u64(ell_plus_one);
  for (auto& H_i : crs_H_vec) vec_H_prime.push_back(g1_to_affine(g1_mul(H_i, beta_inv_pow)));

  std::vector<Fr> vec_b_prime;
  Fr pow_beta = beta;
  for (auto& b_i : vec_b) {
    vec_b_prime.push_back(b_i * pow_beta);
    pow_beta *= beta;
  }
  std::vector<Fr> vec_d, vec_beta_powers;
  pow_beta = Fr::one();
  for (auto& bp : vec_b_prime) {
    vec_d.push_back(bp - pow_beta);
    vec_beta_powers.push_back(pow_beta);
    pow_beta *= beta;
  }
  Fr beta_pow_l1 = beta.pow_u64(ell_plus_one);
  std::vector<Fr> vec_d_blinders;
  for (auto& f : vec_r_b_plus_alpha) vec_d_blinders.push_back(beta_pow_l1 * f);

  std::vector<Fr> vec_alphabeta(n_blinders, alpha * beta_pow_l1);
  G1 D = g1_add(g1_sub(B, g1_msm(vec_G_prime, vec_beta_powers)), g1_msm(vec_H_prime, vec_alphabeta));

  std::vector<G1Aff> vec_G = crs_G_vec;
  vec_G.insert(vec_G.end(), crs_H_vec.begin(), crs_H_vec.end());
  vec_G_prime.insert(vec_G_prime.end(), vec_H_prime.begin(), vec_H_prime.end());

  Fr inner_prod = r_p * beta_pow_l1 + gprod_result * beta.pow_u64(ell) - Fr::one();

  vec_c.insert(vec_c.end(), vec_c_blinders.begin(), vec_c_blinders.end());
  vec_d.insert(vec_d.end(), vec_d_blinders.begin(), vec_d_blinders.end());
  (void)n;

  GrandProductProof pf;
  pf.C = C;
  pf.r_p = r_p;
  pf.ipa_proof = ipa_new(vec_G, vec_G_prime, crs_U, C, D, inner_prod, vec_c, vec_d, transcript, draw);
  return pf;
}

// :180-246
static inline bool gprod_verify(const GrandProductProof& pf, const std::vector<G1Aff>& crs_G_vec, const std::vector<G1Aff>& crs_H_vec,
                                const G1& crs_U, const G1Aff& crs_G_sum, const G1Aff& crs_H_sum, const G1& B, const Fr& gprod_result,
                                size_t n_blinders, Transcript& transcript, MsmAccumulator& acc, const FrDraw& draw) {
  size_t ell = crs_G_vec.size();
  uint64_t ell_plus_one = ell + 1;
  transcript.append_g1("gprod_step1", B);
  transcript.append_fr("gprod_step1", gprod_result);
  Fr alpha = transcript.get_and_append_challenge("gprod_alpha");
  transcript.append_g1("gprod_step2", pf.C);
  transcript.append_fr("gprod_step2", pf.r_p);
  Fr beta = transcript.get_and_append_challenge("gprod_beta");
  Fr beta_inv = beta.inverse();

  std::vector<Fr> vec_u;
  Fr pow_beta_inv = beta_inv;
  for (size_t i = 0; i < ell; i++) {
    vec_u.push_back(pow_beta_inv);
    pow_beta_inv *= beta_inv;
  }
  Fr bl = beta_inv.pow_u64(ell_plus_one);
  for (size_t i = 0; i < n_blinders; i++) vec_u.push_back(bl);

  G1 D = g1_add(g1_sub(B, g1_mul(crs_G_sum, beta_inv)), g1_mul(crs_H_sum, alpha));

  std::vector<G1Aff> vec_G = crs_G_vec;
  vec_G.insert(vec_G.end(), crs_H_vec.begin(), crs_H_vec.end());
  Fr inner_prod = pf.r_p * beta.pow_u64(ell_plus_one) + gprod_result * beta.pow_u64(ell) - Fr::one();
  return ipa_verify(pf.ipa_proof, vec_G, crs_U, pf.C, D, inner_prod, vec_u, transcript, acc, draw);
}

// ---------------------------------------------------------------- same_permutation_argument.rs
struct SamePermutationProof {
  G1 B;
  GrandProductProof grand_product_proof;
};

// :40-101
static inline SamePermutationProof sameperm_new(const std::vector<G1Aff>& crs_G_vec, const std::vector<G1Aff>& crs_H_vec, const G1& crs_U,
                                                const G1& A, const G1& M, const std::vector<Fr>& vec_a,
                                                const std::vector<uint32_t>& permutation, const std::vector<Fr>& vec_a_blinders,
                                                const std::vector<Fr>& vec_m_blinders, Transcript& transcript, const FrDraw& draw) {
  size_t n_blinders = vec_a_blinders.size(), ell = crs_G_vec.size();
  transcript.append_g1("same_perm_step1", A);
  transcript.append_g1("same_perm_step1", M);
  transcript.append_fr_vec("same_perm_step1", vec_a);
  Fr alpha = transcript.get_and_append_challenge("same_perm_alpha");
  Fr beta = transcript.get_and_append_challenge("same_perm_beta");

  std::vector<Fr> vec_a_permuted = get_permutation(vec_a, permutation);
  std::vector<Fr> factors(ell);
  Fr gprod_result = Fr::one();
  for (size_t i = 0; i < ell; i++) {
    factors[i] = vec_a_permuted[i] + Fr::from_u64(permutation[i]) * alpha + beta;
    gprod_result *= factors[i];
  }
  std::vector<Fr> vec_beta_repeated(ell, beta);
  G1 B = g1_add(g1_add(A, g1_mul(M, alpha)), g1_msm(crs_G_vec, vec_beta_repeated));

  std::vector<Fr> vec_b_blinders(n_blinders);
  for (size_t i = 0; i < n_blinders; i++) vec_b_blinders[i] = vec_a_blinders[i] + alpha * vec_m_blinders[i];

  SamePermutationProof pf;
  pf.B = B;
  pf.grand_product_proof = gprod_new(crs_G_vec, crs_H_vec, crs_U, B, gprod_result, factors, vec_b_blinders, transcript, draw);
  return pf;
}

// :112-171
static inline bool sameperm_verify(const SamePermutationProof& pf, const std::vector<G1Aff>& crs_G_vec, const std::vector<G1Aff>& crs_H_vec,
                                   const G1& crs_U, const G1Aff& crs_G_sum, const G1Aff& crs_H_sum, const G1& A, const G1& M,
                                   const std::vector<Fr>& vec_a, size_t n_blinders, Transcript& transcript, MsmAccumulator& acc,
                                   const FrDraw& draw) {
  size_t ell = crs_G_vec.size();
  transcript.append_g1("same_perm_step1", A);
  transcript.append_g1("same_perm_step1", M);
  transcript.append_fr_vec("same_perm_step1", vec_a);
  Fr alpha = transcript.get_and_append_challenge("same_perm_alpha");
  Fr beta = transcript.get_and_append_challenge("same_perm_beta");

  Fr gprod_result = Fr::one();
  size_t m = std::min(ell, vec_a.size());   // Rust `zip`
  for (size_t i = 0; i < m; i++) gprod_result *= vec_a[i] + Fr::from_u64(i) * alpha + beta;

  std::vector<Fr> vec_beta_repeated(ell, beta);
  acc.accumulate_check(g1_sub(g1_sub(pf.B, A), g1_mul(M, alpha)), vec_beta_repeated, crs_G_vec, draw);
  return gprod_verify(pf.grand_product_proof, crs_G_vec, crs_H_vec, crs_U, crs_G_sum, crs_H_sum, pf.B, gprod_result, n_blinders,
                      transcript, acc, draw);
}

// ---------------------------------------------------------------- same_scalar_argument.rs
struct SameScalarProof {
  GroupCommitment cm_A, cm_B;
  Fr z_k, z_t, z_u;
};
static inline void samescalar_transcript(Transcript& t, const G1& R, const G1& S, const GroupCommitment& cm_T, const GroupCommitment& cm_U,
                                         const GroupCommitment& cm_A, const GroupCommitment& cm_B) {   // :63-69, :112-126
  const G1* pts[10] = {&R, &S, &cm_T.T_1, &cm_T.T_2, &cm_U.T_1, &cm_U.T_2, &cm_A.T_1, &cm_A.T_2, &cm_B.T_1, &cm_B.T_2};
  for (auto p : pts) t.append_g1("sameexp_points", *p);
}
// :39-84
static inline SameScalarProof samescalar_new(const G1& crs_G_t, const G1& crs_G_u, const G1& crs_H, const G1& R, const G1& S,
                                             const GroupCommitment& cm_T, const GroupCommitment& cm_U, const Fr& k, const Fr& r_t,
                                             const Fr& r_u, Transcript& transcript, const FrDraw& draw) {
  Fr r_a = draw(), r_b = draw(), r_k = draw();
  SameScalarProof pf;
  pf.cm_A = GroupCommitment::make(crs_G_t, crs_H, g1_mul(R, r_k), r_a);
  pf.cm_B = GroupCommitment::make(crs_G_u, crs_H, g1_mul(S, r_k), r_b);
  samescalar_transcript(transcript, R, S, cm_T, cm_U, pf.cm_A, pf.cm_B);
  Fr alpha = transcript.get_and_append_challenge("same_scalar_alpha");
  pf.z_k = r_k + k * alpha;
  pf.z_t = r_a + r_t * alpha;
  pf.z_u = r_b + r_u * alpha;
  return pf;
}
// :96-137
static inline bool samescalar_verify(const SameScalarProof& pf, const G1& crs_G_t, const G1& crs_G_u, const G1& crs_H, const G1& R,
                                     const G1& S, const GroupCommitment& cm_T, const GroupCommitment& cm_U, Transcript& transcript) {
  samescalar_transcript(transcript, R, S, cm_T, cm_U, pf.cm_A, pf.cm_B);
  Fr alpha = transcript.get_and_append_challenge("same_scalar_alpha");
  GroupCommitment expected_1 = GroupCommitment::make(crs_G_t, crs_H, g1_mul(R, pf.z_k), pf.z_t);
  GroupCommitment expected_2 = GroupCommitment::make(crs_G_u, crs_H, g1_mul(S, pf.z_k), pf.z_u);
  return pf.cm_A.add(cm_T.mul(alpha)).eq(expected_1) && pf.cm_B.add(cm_U.mul(alpha)).eq(expected_2);
}

// ---------------------------------------------------------------- same_multiscalar_argument.rs
struct SameMultiscalarProof {
  G1 B_a, B_t, B_u;
  std::vector<G1> vec_L_A, vec_L_T, vec_L_U, vec_R_A, vec_R_T, vec_R_U;
  Fr x_final;
};

// :54-150
static inline SameMultiscalarProof samemsm_new(std::vector<G1Aff> crs_G_vec, const G1& A, const G1& Z_t, const G1& Z_u,
                                               std::vector<G1Aff> vec_T, std::vector<G1Aff> vec_U, std::vector<Fr> vec_x,
                                               Transcript& transcript, const FrDraw& draw) {
  size_t n = vec_x.size();
  SameMultiscalarProof pf;
  std::vector<Fr> vec_r = generate_blinders(draw, n);
  pf.B_a = g1_msm(crs_G_vec, vec_r);
  pf.B_t = g1_msm(vec_T, vec_r);
  pf.B_u = g1_msm(vec_U, vec_r);

  transcript.append_g1("same_msm_step1", A);
  transcript.append_g1("same_msm_step1", Z_t);
  transcript.append_g1("same_msm_step1", Z_u);
  transcript.append_g1_vec("same_msm_step1", vec_T);
  transcript.append_g1_vec("same_msm_step1", vec_U);
  transcript.append_g1("same_msm_step1", pf.B_a);
  transcript.append_g1("same_msm_step1", pf.B_t);
  transcript.append_g1("same_msm_step1", pf.B_u);
  Fr alpha = transcript.get_and_append_challenge("same_msm_alpha");

  for (size_t i = 0; i < n; i++) vec_x[i] = vec_r[i] + alpha * vec_x[i];

  while (n > 1) {
    n /= 2;
    Fr* x_L = vec_x.data();
    Fr* x_R = x_L + n;
    G1Aff *T_L = vec_T.data(), *T_R = T_L + n, *U_L = vec_U.data(), *U_R = U_L + n, *G_L = crs_G_vec.data(), *G_R = G_L + n;

    G1 L_A = g1_msm(G_R, x_L, n), L_T = g1_msm(T_R, x_L, n), L_U = g1_msm(U_R, x_L, n);
    G1 R_A = g1_msm(G_L, x_R, n), R_T = g1_msm(T_L, x_R, n), R_U = g1_msm(U_L, x_R, n);
    pf.vec_L_A.push_back(L_A);
    pf.vec_L_T.push_back(L_T);
    pf.vec_L_U.push_back(L_U);
    pf.vec_R_A.push_back(R_A);
    pf.vec_R_T.push_back(R_T);
    pf.vec_R_U.push_back(R_U);

    const G1* pts[6] = {&L_A, &L_T, &L_U, &R_A, &R_T, &R_U};
    for (auto p : pts) transcript.append_g1("same_msm_loop", *p);
    Fr gamma = transcript.get_and_append_challenge("same_msm_gamma");
    Fr gamma_inv = gamma.inverse();

    for (size_t i = 0; i < n; i++) {
      x_L[i] += gamma_inv * x_R[i];
      T_L[i] = g1_to_affine(g1_add(G1::from_affine(T_L[i]), g1_mul(T_R[i], gamma)));
      U_L[i] = g1_to_affine(g1_add(G1::from_affine(U_L[i]), g1_mul(U_R[i], gamma)));
      G_L[i] = g1_to_affine(g1_add(G1::from_affine(G_L[i]), g1_mul(G_R[i], gamma)));
    }
  }
  pf.x_final = vec_x[0];
  return pf;
}

// :153-261
static inline bool samemsm_verify(const SameMultiscalarProof& pf, const std::vector<G1Aff>& crs_G_vec, const G1& A, const G1& Z_t,
                                  const G1& Z_u, const std::vector<G1Aff>& vec_T, const std::vector<G1Aff>& vec_U, Transcript& transcript,
                                  MsmAccumulator& acc, const FrDraw& draw) {
  size_t n = vec_T.size();
  transcript.append_g1("same_msm_step1", A);
  transcript.append_g1("same_msm_step1", Z_t);
  transcript.append_g1("same_msm_step1", Z_u);
  transcript.append_g1_vec("same_msm_step1", vec_T);
  transcript.append_g1_vec("same_msm_step1", vec_U);
  transcript.append_g1("same_msm_step1", pf.B_a);
  transcript.append_g1("same_msm_step1", pf.B_t);
  transcript.append_g1("same_msm_step1", pf.B_u);
  Fr alpha = transcript.get_and_append_challenge("same_msm_alpha");

  size_t lg_n = pf.vec_L_A.size();
  if (lg_n >= 32) return false;
  if (n != (size_t(1) << lg_n)) return false;
  auto bits = get_verification_scalars_bitstring(n, lg_n);
  std::vector<Fr> vec_gamma;
  for (size_t i = 0; i < lg_n; i++) {
    const G1* pts[6] = {&pf.vec_L_A[i], &pf.vec_L_T[i], &pf.vec_L_U[i], &pf.vec_R_A[i], &pf.vec_R_T[i], &pf.vec_R_U[i]};
    for (auto p : pts) transcript.append_g1("same_msm_loop", *p);
    vec_gamma.push_back(transcript.get_and_append_challenge("same_msm_gamma"));
  }
  std::vector<Fr> vec_gamma_inv = vec_gamma;
  batch_inversion(vec_gamma_inv);
  std::vector<Fr> vec_x_times_s(n);
  for (size_t i = 0; i < n; i++) {
    Fr s = Fr::one();
    for (size_t j : bits[i]) s *= vec_gamma[j];
    vec_x_times_s[i] = pf.x_final * s;
  }

  G1 A_a = g1_add(pf.B_a, g1_mul(A, alpha));
  G1 Z_t_a = g1_add(pf.B_t, g1_mul(Z_t, alpha));
  G1 Z_u_a = g1_add(pf.B_u, g1_mul(Z_u, alpha));

  G1 lhs = g1_add(g1_add(g1_msm_from_projective(pf.vec_L_A, vec_gamma), A_a), g1_msm_from_projective(pf.vec_R_A, vec_gamma_inv));
  acc.accumulate_check(lhs, vec_x_times_s, crs_G_vec, draw);
  lhs = g1_add(g1_add(g1_msm_from_projective(pf.vec_L_T, vec_gamma), Z_t_a), g1_msm_from_projective(pf.vec_R_T, vec_gamma_inv));
  acc.accumulate_check(lhs, vec_x_times_s, vec_T, draw);
  lhs = g1_add(g1_add(g1_msm_from_projective(pf.vec_L_U, vec_gamma), Z_u_a), g1_msm_from_projective(pf.vec_R_U, vec_gamma_inv));
  acc.accumulate_check(lhs, vec_x_times_s, vec_U, draw);
  return true;
}

// ---------------------------------------------------------------- curdleproofs.rs
struct CurdleproofsProof {
  G1 A;
  GroupCommitment cm_T, cm_U;
  G1 R, S;
  SamePermutationProof same_perm_proof;
  SameScalarProof same_scalar_proof;
  SameMultiscalarProof same_multiscalar_proof;
};

// util.rs:83-106
static inline void shuffle_permute_and_commit_input(const Crs& crs, const std::vector<G1Aff>& vec_R, const std::vector<G1Aff>& vec_S,
                                                    const std::vector<uint32_t>& permutation, const Fr& k, const FrDraw& draw,
                                                    std::vector<G1Aff>* vec_T, std::vector<G1Aff>* vec_U, G1* M,
                                                    std::vector<Fr>* vec_m_blinders) {
  size_t ell = crs.vec_G.size();
  std::vector<G1Aff> T(vec_R.size()), U(vec_S.size());
  for (size_t i = 0; i < vec_R.size(); i++) T[i] = g1_to_affine(g1_mul(vec_R[i], k));
  for (size_t i = 0; i < vec_S.size(); i++) U[i] = g1_to_affine(g1_mul(vec_S[i], k));
  *vec_T = get_permutation(T, permutation);
  *vec_U = get_permutation(U, permutation);
  std::vector<Fr> range_as_fr(ell);
  for (size_t i = 0; i < ell; i++) range_as_fr[i] = Fr::from_u64(i);
  std::vector<Fr> sigma_ell = get_permutation(range_as_fr, permutation);
  *vec_m_blinders = generate_blinders(draw, N_BLINDERS);
  *M = g1_add(g1_msm(crs.vec_G, sigma_ell), g1_msm(crs.vec_H, *vec_m_blinders));
}

static inline void with_blinders(const Crs& crs, const std::vector<G1Aff>& vec_T, const std::vector<G1Aff>& vec_U,
                                 std::vector<G1Aff>* G_b, std::vector<G1Aff>* T_b, std::vector<G1Aff>* U_b) {   // curdleproofs.rs:136-155, 260-279
  *G_b = crs.vec_G;
  G_b->insert(G_b->end(), crs.vec_H.begin(), crs.vec_H.begin() + (N_BLINDERS - 2));
  G_b->push_back(g1_to_affine(crs.G_t));
  G_b->push_back(g1_to_affine(crs.G_u));
  G1Aff Haff = g1_to_affine(crs.H), O = G1Aff::identity();
  *T_b = vec_T;
  T_b->insert(T_b->end(), {O, O, Haff, O});
  *U_b = vec_U;
  U_b->insert(U_b->end(), {O, O, O, Haff});
}

// curdleproofs.rs:59-184
static inline CurdleproofsProof curdleproofs_new(const Crs& crs, const std::vector<G1Aff>& vec_R, const std::vector<G1Aff>& vec_S,
                                                 const std::vector<G1Aff>& vec_T, const std::vector<G1Aff>& vec_U, const G1& M,
                                                 const std::vector<uint32_t>& permutation, const Fr& k,
                                                 const std::vector<Fr>& vec_m_blinders, const FrDraw& draw) {
  size_t ell = vec_R.size();
  Transcript transcript("curdleproofs");
  transcript.append_g1_vec("curdleproofs_step1", vec_R);
  transcript.append_g1_vec("curdleproofs_step1", vec_S);
  transcript.append_g1_vec("curdleproofs_step1", vec_T);
  transcript.append_g1_vec("curdleproofs_step1", vec_U);
  transcript.append_g1("curdleproofs_step1", M);
  std::vector<Fr> vec_a = transcript.get_and_append_challenges("curdleproofs_vec_a", ell);

  std::vector<Fr> vec_a_blinders = generate_blinders(draw, N_BLINDERS - 2);
  std::vector<Fr> vec_r_a_prime = vec_a_blinders;
  vec_r_a_prime.push_back(Fr::zero());
  vec_r_a_prime.push_back(Fr::zero());
  std::vector<Fr> vec_a_permuted = get_permutation(vec_a, permutation);

  CurdleproofsProof pf;
  pf.A = g1_add(g1_msm(crs.vec_G, vec_a_permuted), g1_msm(crs.vec_H, vec_r_a_prime));
  pf.same_perm_proof = sameperm_new(crs.vec_G, crs.vec_H, crs.H, pf.A, M, vec_a, permutation, vec_r_a_prime, vec_m_blinders, transcript, draw);

  Fr r_t = draw(), r_u = draw();
  pf.R = g1_msm(vec_R, vec_a);
  pf.S = g1_msm(vec_S, vec_a);
  pf.cm_T = GroupCommitment::make(crs.G_t, crs.H, g1_mul(pf.R, k), r_t);
  pf.cm_U = GroupCommitment::make(crs.G_u, crs.H, g1_mul(pf.S, k), r_u);
  pf.same_scalar_proof = samescalar_new(crs.G_t, crs.G_u, crs.H, pf.R, pf.S, pf.cm_T, pf.cm_U, k, r_t, r_u, transcript, draw);

  G1 A_prime = g1_add(g1_add(pf.A, pf.cm_T.T_1), pf.cm_U.T_1);
  std::vector<G1Aff> G_b, T_b, U_b;
  with_blinders(crs, vec_T, vec_U, &G_b, &T_b, &U_b);
  std::vector<Fr> vec_a_with_blinders = vec_a_permuted;
  vec_a_with_blinders.insert(vec_a_with_blinders.end(), vec_a_blinders.begin(), vec_a_blinders.end());
  vec_a_with_blinders.push_back(r_t);
  vec_a_with_blinders.push_back(r_u);
  pf.same_multiscalar_proof = samemsm_new(G_b, A_prime, pf.cm_T.T_2, pf.cm_U.T_2, T_b, U_b, vec_a_with_blinders, transcript, draw);
  return pf;
}

// curdleproofs.rs:197-298
static inline bool curdleproofs_verify(const CurdleproofsProof& pf, const Crs& crs, const std::vector<G1Aff>& vec_R,
                                       const std::vector<G1Aff>& vec_S, const std::vector<G1Aff>& vec_T, const std::vector<G1Aff>& vec_U,
                                       const G1& M, const FrDraw& draw) {
  size_t ell = vec_R.size();
  Transcript transcript("curdleproofs");
  MsmAccumulator acc;
  if (vec_T.empty() || vec_T[0].inf) return false;   // :218 (an empty vec_T would panic in Rust)
  transcript.append_g1_vec("curdleproofs_step1", vec_R);
  transcript.append_g1_vec("curdleproofs_step1", vec_S);
  transcript.append_g1_vec("curdleproofs_step1", vec_T);
  transcript.append_g1_vec("curdleproofs_step1", vec_U);
  transcript.append_g1("curdleproofs_step1", M);
  std::vector<Fr> vec_a = transcript.get_and_append_challenges("curdleproofs_vec_a", ell);

  if (!sameperm_verify(pf.same_perm_proof, crs.vec_G, crs.vec_H, crs.H, crs.G_sum, crs.H_sum, pf.A, M, vec_a, N_BLINDERS, transcript, acc, draw))
    return false;
  if (!samescalar_verify(pf.same_scalar_proof, crs.G_t, crs.G_u, crs.H, pf.R, pf.S, pf.cm_T, pf.cm_U, transcript)) return false;

  G1 A_prime = g1_add(g1_add(pf.A, pf.cm_T.T_1), pf.cm_U.T_1);
  std::vector<G1Aff> G_b, T_b, U_b;
  with_blinders(crs, vec_T, vec_U, &G_b, &T_b, &U_b);
  if (!samemsm_verify(pf.same_multiscalar_proof, G_b, A_prime, pf.cm_T.T_2, pf.cm_U.T_2, T_b, U_b, transcript, acc, draw)) return false;

  acc.accumulate_check(pf.R, vec_a, vec_R, draw);
  acc.accumulate_check(pf.S, vec_a, vec_S, draw);
  return acc.verify();
}

// ---------------------------------------------------------------- (de)serialisation
struct ByteWriter {
  std::vector<uint8_t> b;
  void g1(const G1& p) {
    uint8_t t[48];
    g1_compress(p, t);
    b.insert(b.end(), t, t + 48);
  }
  void fr(const Fr& x) {
    uint8_t t[32];
    fr_to_bytes(x, t);
    b.insert(b.end(), t, t + 32);
  }
  void g1v(const std::vector<G1>& v) {   // util.rs:126-133 (no length prefix)
    for (auto& p : v) g1(p);
  }
};
struct ByteReader {
  const uint8_t* p;
  size_t left;
  bool ok = true;
  G1 g1() {
    G1Aff a = G1Aff::identity();
    if (left < 48 || !g1_decompress(p, &a)) {
      ok = false;
      return G1::identity();
    }
    p += 48;
    left -= 48;
    return G1::from_affine(a);
  }
  Fr fr() {
    Fr x = Fr::zero();
    if (left < 32 || !fr_from_bytes(p, &x)) {
      ok = false;
      return x;
    }
    p += 32;
    left -= 32;
    return x;
  }
  std::vector<G1> g1v(size_t n) {   // util.rs:116-124
    std::vector<G1> v;
    for (size_t i = 0; i < n && ok; i++) v.push_back(g1());
    return v;
  }
};

// curdleproofs.rs:300-310 with the nested serialisers (same_permutation_argument.rs:173-177,
// grand_product_argument.rs:248-253, inner_product_argument.rs:328-338, same_scalar_argument.rs:17-23 derive,
// same_multiscalar_argument.rs:263-275)
static inline std::vector<uint8_t> curdleproofs_serialize(const CurdleproofsProof& pf) {
  ByteWriter w;
  w.g1(pf.A);
  w.g1(pf.cm_T.T_1);
  w.g1(pf.cm_T.T_2);
  w.g1(pf.cm_U.T_1);
  w.g1(pf.cm_U.T_2);
  w.g1(pf.R);
  w.g1(pf.S);
  w.g1(pf.same_perm_proof.B);
  const GrandProductProof& gp = pf.same_perm_proof.grand_product_proof;
  w.g1(gp.C);
  w.fr(gp.r_p);
  const InnerProductProof& ip = gp.ipa_proof;
  w.g1(ip.B_c);
  w.g1(ip.B_d);
  w.g1v(ip.vec_L_C);
  w.g1v(ip.vec_R_C);
  w.g1v(ip.vec_L_D);
  w.g1v(ip.vec_R_D);
  w.fr(ip.c_final);
  w.fr(ip.d_final);
  const SameScalarProof& ss = pf.same_scalar_proof;
  w.g1(ss.cm_A.T_1);
  w.g1(ss.cm_A.T_2);
  w.g1(ss.cm_B.T_1);
  w.g1(ss.cm_B.T_2);
  w.fr(ss.z_k);
  w.fr(ss.z_t);
  w.fr(ss.z_u);
  const SameMultiscalarProof& sm = pf.same_multiscalar_proof;
  w.g1(sm.B_a);
  w.g1(sm.B_t);
  w.g1(sm.B_u);
  w.g1v(sm.vec_L_A);
  w.g1v(sm.vec_L_T);
  w.g1v(sm.vec_L_U);
  w.g1v(sm.vec_R_A);
  w.g1v(sm.vec_R_T);
  w.g1v(sm.vec_R_U);
  w.fr(sm.x_final);
  return w.b;
}
// curdleproofs.rs:312-323
static inline bool curdleproofs_deserialize(const uint8_t* bytes, size_t len, size_t log2_n, CurdleproofsProof* pf) {
  ByteReader r{bytes, len};
  pf->A = r.g1();
  pf->cm_T.T_1 = r.g1();
  pf->cm_T.T_2 = r.g1();
  pf->cm_U.T_1 = r.g1();
  pf->cm_U.T_2 = r.g1();
  pf->R = r.g1();
  pf->S = r.g1();
  pf->same_perm_proof.B = r.g1();
  GrandProductProof& gp = pf->same_perm_proof.grand_product_proof;
  gp.C = r.g1();
  gp.r_p = r.fr();
  InnerProductProof& ip = gp.ipa_proof;
  ip.B_c = r.g1();
  ip.B_d = r.g1();
  ip.vec_L_C = r.g1v(log2_n);
  ip.vec_R_C = r.g1v(log2_n);
  ip.vec_L_D = r.g1v(log2_n);
  ip.vec_R_D = r.g1v(log2_n);
  ip.c_final = r.fr();
  ip.d_final = r.fr();
  SameScalarProof& ss = pf->same_scalar_proof;
  ss.cm_A.T_1 = r.g1();
  ss.cm_A.T_2 = r.g1();
  ss.cm_B.T_1 = r.g1();
  ss.cm_B.T_2 = r.g1();
  ss.z_k = r.fr();
  ss.z_t = r.fr();
  ss.z_u = r.fr();
  SameMultiscalarProof& sm = pf->same_multiscalar_proof;
  sm.B_a = r.g1();
  sm.B_t = r.g1();
  sm.B_u = r.g1();
  sm.vec_L_A = r.g1v(log2_n);
  sm.vec_L_T = r.g1v(log2_n);
  sm.vec_L_U = r.g1v(log2_n);
  sm.vec_R_A = r.g1v(log2_n);
  sm.vec_R_T = r.g1v(log2_n);
  sm.vec_R_U = r.g1v(log2_n);
  sm.x_final = r.fr();
  return r.ok;
}

// ---------------------------------------------------------------- whisk.rs
struct WhiskTracker {
  uint8_t r_G[48], k_r_G[48];
};
static inline WhiskTracker tracker_from_k_r(const Fr& k, const Fr& r) {   // whisk.rs:45-55
  G1 r_G = g1_mul(g1_generator(), r);
  G1 k_r_G = g1_mul(g1_to_affine(r_G), k);
  WhiskTracker t;
  g1_compress(r_G, t.r_G);
  g1_compress(k_r_G, t.k_r_G);
  return t;
}
// whisk.rs:228-263
static inline bool generate_whisk_tracker_proof(const FrDraw& draw, const WhiskTracker& tracker, const Fr& k, uint8_t out[128]) {
  G1Aff k_r_g, r_g, G = g1_generator();
  if (!g1_decompress(tracker.k_r_G, &k_r_g) || !g1_decompress(tracker.r_G, &r_g)) return false;
  G1 k_G = g1_mul(G, k);
  Fr blinder = draw();
  G1 A = g1_mul(G, blinder), B = g1_mul(r_g, blinder);
  Transcript transcript("whisk_opening_proof");
  transcript.append_g1("tracker_opening_proof", k_G);
  transcript.append_g1("tracker_opening_proof", G);
  transcript.append_g1("tracker_opening_proof", k_r_g);
  transcript.append_g1("tracker_opening_proof", r_g);
  transcript.append_g1("tracker_opening_proof", A);
  transcript.append_g1("tracker_opening_proof", B);
  Fr challenge = transcript.get_and_append_challenge("tracker_opening_proof_challenge");
  Fr s = blinder - challenge * k;
  g1_compress(A, out);
  g1_compress(B, out + 48);
  fr_to_bytes(s, out + 96);
  return true;
}
// whisk.rs:183-226
static inline int is_valid_whisk_tracker_proof(const WhiskTracker& tracker, const uint8_t k_commitment[48], const uint8_t proof[128]) {
  G1Aff A, B, k_r_G, r_G, k_G, G = g1_generator();
  Fr s;
  if (!g1_decompress(proof, &A) || !g1_decompress(proof + 48, &B) || !fr_from_bytes(proof + 96, &s)) return -1;
  if (!g1_decompress(tracker.k_r_G, &k_r_G) || !g1_decompress(tracker.r_G, &r_G) || !g1_decompress(k_commitment, &k_G)) return -1;
  Transcript transcript("whisk_opening_proof");
  transcript.append_g1("tracker_opening_proof", k_G);
  transcript.append_g1("tracker_opening_proof", G);
  transcript.append_g1("tracker_opening_proof", k_r_G);
  transcript.append_g1("tracker_opening_proof", r_G);
  transcript.append_g1("tracker_opening_proof", A);
  transcript.append_g1("tracker_opening_proof", B);
  Fr challenge = transcript.get_and_append_challenge("tracker_opening_proof_challenge");
  G1 A_prime = g1_add(g1_mul(G, s), g1_mul(k_G, challenge));
  G1 B_prime = g1_add(g1_mul(r_G, s), g1_mul(k_r_G, challenge));
  return (g1_eq(A_prime, G1::from_affine(A)) && g1_eq(B_prime, G1::from_affine(B))) ? 1 : 0;
}

static inline bool unzip_trackers(const std::vector<WhiskTracker>& tr, std::vector<G1Aff>* r, std::vector<G1Aff>* s) {   // whisk.rs:265-277
  r->resize(tr.size());
  s->resize(tr.size());
  for (size_t i = 0; i < tr.size(); i++)
    if (!g1_decompress(tr[i].r_G, &(*r)[i])) return false;
  for (size_t i = 0; i < tr.size(); i++)
    if (!g1_decompress(tr[i].k_r_G, &(*s)[i])) return false;
  return true;
}

// whisk.rs:144-179 (ell = crs.vec_G.len() here; the reference hard-codes N = 128)
// `rng` supplies both the u32 stream for the permutation shuffle and the Fr draws.
static inline bool generate_whisk_shuffle_proof(StdRng& rng, const Crs& crs, const std::vector<WhiskTracker>& pre_trackers,
                                                std::vector<WhiskTracker>* post_trackers, std::vector<uint8_t>* proof_bytes) {
  size_t ell = crs.vec_G.size();
  std::vector<uint32_t> permutation(ell);
  for (size_t i = 0; i < ell; i++) permutation[i] = (uint32_t)i;
  shuffle(rng, permutation);
  FrDraw draw = [&rng]() { return rand_fr(rng); };
  Fr k = draw();
  std::vector<G1Aff> vec_r, vec_s, vec_t, vec_u;
  if (!unzip_trackers(pre_trackers, &vec_r, &vec_s)) return false;
  G1 m;
  std::vector<Fr> vec_m_blinders;
  shuffle_permute_and_commit_input(crs, vec_r, vec_s, permutation, k, draw, &vec_t, &vec_u, &m, &vec_m_blinders);
  CurdleproofsProof proof = curdleproofs_new(crs, vec_r, vec_s, vec_t, vec_u, m, permutation, k, vec_m_blinders, draw);
  ByteWriter w;
  w.g1(m);
  std::vector<uint8_t> body = curdleproofs_serialize(proof);
  w.b.insert(w.b.end(), body.begin(), body.end());
  *proof_bytes = w.b;
  post_trackers->resize(ell);
  for (size_t i = 0; i < ell; i++) {
    g1_compress(vec_t[i], (*post_trackers)[i].r_G);
    g1_compress(vec_u[i], (*post_trackers)[i].k_r_G);
  }
  return true;
}

// whisk.rs:106-130; returns 1 valid, 0 invalid, -1 deserialisation error
static inline int is_valid_whisk_shuffle_proof(const FrDraw& draw, const Crs& crs, const std::vector<WhiskTracker>& pre,
                                               const std::vector<WhiskTracker>& post, const uint8_t* proof_bytes, size_t len) {
  std::vector<G1Aff> vec_r, vec_s, vec_t, vec_u;
  if (!unzip_trackers(pre, &vec_r, &vec_s) || !unzip_trackers(post, &vec_t, &vec_u)) return -1;
  if (len < 48) return -1;
  G1Aff M;
  if (!g1_decompress(proof_bytes, &M)) return -1;
  CurdleproofsProof pf;
  if (!curdleproofs_deserialize(proof_bytes + 48, len - 48, crs.log2_n(), &pf)) return -1;
  return curdleproofs_verify(pf, crs, vec_r, vec_s, vec_t, vec_u, G1::from_affine(M), draw) ? 1 : 0;
}

}  // namespace orc
