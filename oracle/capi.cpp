// ORACLE (test infrastructure — never linked into the product path).
//
// C entry points over the CPU restatement (protocol.h et al.) so that tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() can drive it through ctypes.
// Buffer layouts are the ones the product C-ABI (include/cpx.h) uses, i.e. arkworks' in-memory
// limbs: Fp = 6 x u64 LE Montgomery (R = 2^384), Fr = 4 x u64 LE Montgomery (R = 2^256);
//   affine point   96 B  = x || y           (identity = 96 zero bytes; (0,0) is not on the curve)
//   Jacobian point 144 B = X || Y || Z      (identity = Z == 0)
#include "protocol.h"
#include "sha256.h"
#include <chrono>

using namespace orc;

namespace {

G1Aff aff_from_wire(const uint8_t* b) {
  G1Aff p;
  memcpy(p.x.v, b, 48);
  memcpy(p.y.v, b + 48, 48);
  p.inf = p.x.is_zero() && p.y.is_zero();
  return p;
}
void aff_to_wire(const G1Aff& p, uint8_t* b) {
  if (p.inf) {
    memset(b, 0, 96);
    return;
  }
  memcpy(b, p.x.v, 48);
  memcpy(b + 48, p.y.v, 48);
}
G1 jac_from_wire(const uint8_t* b) {
  G1 p;
  memcpy(p.x.v, b, 48);
  memcpy(p.y.v, b + 48, 48);
  memcpy(p.z.v, b + 96, 48);
  return p;
}
void jac_to_wire(const G1& p, uint8_t* b) {
  memcpy(b, p.x.v, 48);
  memcpy(b + 48, p.y.v, 48);
  memcpy(b + 96, p.z.v, 48);
}
Fr fr_from_wire(const uint8_t* b) {
  Fr x;
  memcpy(x.v, b, 32);
  return x;
}
void fr_to_wire(const Fr& x, uint8_t* b) { memcpy(b, x.v, 32); }
std::vector<G1Aff> affs_from_wire(const uint8_t* b, size_t n) {
  std::vector<G1Aff> v(n);
  for (size_t i = 0; i < n; i++) v[i] = aff_from_wire(b + 96 * i);
  return v;
}
std::vector<Fr> frs_from_wire(const uint8_t* b, size_t n) {
  std::vector<Fr> v(n);
  for (size_t i = 0; i < n; i++) v[i] = fr_from_wire(b + 32 * i);
  return v;
}

struct Recorder {   // draws from an StdRng and remembers what was drawn
  StdRng* rng;
  std::vector<Fr> log;
  Fr operator()() {
    Fr x = rand_fr(*rng);
    log.push_back(x);
    return x;
  }
};
struct Replayer {   // replays caller-supplied randomness in draw order
  const uint8_t* buf;
  size_t n, pos = 0;
  bool overrun = false;
  Fr operator()() {
    if (pos >= n) {
      overrun = true;
      return Fr::one();
    }
    return fr_from_wire(buf + 32 * pos++);
  }
};

bool crs_from_wire(size_t ell, const uint8_t* crs_points, Crs* crs) {
  return crs_from_points(ell, affs_from_wire(crs_points, ell + N_BLINDERS + CRS_EXTRA_POINTS), crs);
}

}  // namespace

extern "C" {

// ---- self checks against published / reference-held vectors ----
void orc_keccak_f1600(uint64_t st[25]) { keccak_f1600(st); }

// merlin's own test vector (merlin 3.0.0 transcript.rs test `equivalence_simple`)
void orc_merlin_test_vector(uint8_t out[32]) {
  Transcript t("test protocol");
  t.append_message("some label", (const uint8_t*)"some data", 9);
  t.challenge_bytes("challenge", out, 32);
}

// transcript.rs:41-54 on a small transcript: hash `len` bytes under "lbl", return two challenges "ch"
void orc_challenges(const uint8_t* msg, size_t len, uint8_t out[64]) {
  Transcript t("curdleproofs");
  t.append_message("lbl", msg, len);
  fr_to_wire(t.get_and_append_challenge("ch"), out);
  fr_to_wire(t.get_and_append_challenge("ch"), out + 32);
}

// whisk.rs:381-402 `whisk_tracker_proof`: returns validity, writes the 128-byte proof
int orc_kat_tracker_proof(uint8_t out[128]) {
  StdRng rng(0);
  FrDraw draw = [&rng]() { return rand_fr(rng); };
  Fr k = draw();
  WhiskTracker tracker = tracker_from_k_r(k, draw());   // from_k: r = Fr::rand
  uint8_t k_commitment[48];
  g1_compress(g1_mul(g1_generator(), k), k_commitment);
  if (!generate_whisk_tracker_proof(draw, tracker, k, out)) return -1;
  return is_valid_whisk_tracker_proof(tracker, k_commitment, out);
}

// whisk.rs:416-456 `whisk_shuffle_proof` (ell = 124 there). Writes 48*(2*ell) bytes of pre and
// post trackers (r_G || k_r_G per tracker) if the pointers are non-null, and the proof bytes.
int orc_kat_shuffle_proof(size_t ell, uint8_t* pre_out, uint8_t* post_out, uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
  StdRng rng(0);
  Crs crs = generate_crs(ell);
  std::vector<WhiskTracker> pre(ell), post;
  for (auto& t : pre) {   // generate_shuffle_trackers -> from_rand: k = rand, then from_k draws r
    Fr k = rand_fr(rng);
    Fr r = rand_fr(rng);
    t = tracker_from_k_r(k, r);
  }
  std::vector<uint8_t> proof;
  if (!generate_whisk_shuffle_proof(rng, crs, pre, &post, &proof)) return -1;
  FrDraw draw = [&rng]() { return rand_fr(rng); };
  int valid = is_valid_whisk_shuffle_proof(draw, crs, pre, post, proof.data(), proof.size());
  if (proof.size() > proof_cap) return -2;
  memcpy(proof_out, proof.data(), proof.size());
  *proof_len = proof.size();
  if (pre_out) memcpy(pre_out, pre.data(), 96 * ell);
  if (post_out) memcpy(post_out, post.data(), 96 * ell);
  return valid;
}

// ---- RNG model ----
void* orc_rng_new(uint64_t seed) { return new StdRng(seed); }
void orc_rng_free(void* r) { delete (StdRng*)r; }
uint32_t orc_rng_u32(void* r) { return ((StdRng*)r)->next_u32(); }
void orc_rng_key(void* r, uint8_t out[32]) { memcpy(out, ((StdRng*)r)->key, 32); }
void orc_rng_fr(void* r, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) fr_to_wire(rand_fr(*(StdRng*)r), out + 32 * i);
}
void orc_rng_g1_affine(void* r, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) aff_to_wire(g1_to_affine(rand_g1(*(StdRng*)r)), out + 96 * i);
}
void orc_rng_shuffle(void* r, uint32_t* perm, size_t n) {
  std::vector<uint32_t> p(perm, perm + n);
  shuffle(*(StdRng*)r, p);
  memcpy(perm, p.data(), 4 * n);
}

// ---- field / group primitives (kernel-level parity checks) ----
void orc_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Fp x, y;
    memcpy(x.v, a + 48 * i, 48);
    memcpy(y.v, b + 48 * i, 48);
    Fp z = x * y;
    memcpy(out + 48 * i, z.v, 48);
  }
}
void orc_fp_inv(const uint8_t* a, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Fp x;
    memcpy(x.v, a + 48 * i, 48);
    Fp z = x.inverse();
    memcpy(out + 48 * i, z.v, 48);
  }
}
void orc_fr_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) fr_to_wire(fr_from_wire(a + 32 * i) * fr_from_wire(b + 32 * i), out + 32 * i);
}
void orc_fr_inv(const uint8_t* a, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) fr_to_wire(fr_from_wire(a + 32 * i).inverse(), out + 32 * i);
}
void orc_fr_from_u64(uint64_t x, uint8_t out[32]) { fr_to_wire(Fr::from_u64(x), out); }
void orc_fr_to_canonical_bytes(const uint8_t* a, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) fr_to_bytes(fr_from_wire(a + 32 * i), out + 32 * i);
}
int orc_fr_from_canonical_bytes(const uint8_t* a, uint8_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Fr x;
    if (!fr_from_bytes(a + 32 * i, &x)) return -1;
    fr_to_wire(x, out + 32 * i);
  }
  return 0;
}
void orc_fr_from_le_bytes_mod_order(const uint8_t* a, size_t len, uint8_t out[32]) { fr_to_wire(fr_from_le_bytes_mod_order(a, len), out); }

void orc_g1_generator(uint8_t out[96]) { aff_to_wire(g1_generator(), out); }
int orc_g1_on_curve(const uint8_t* p) { return g1_on_curve(aff_from_wire(p)) ? 1 : 0; }
int orc_g1_in_subgroup(const uint8_t* p) { return g1_in_subgroup(aff_from_wire(p)) ? 1 : 0; }

// util.rs:19-22
void orc_g1_msm(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[144]) {
  jac_to_wire(g1_msm(affs_from_wire(bases, n), frs_from_wire(scalars, n)), out);
}
void orc_g1_msm_naive(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[144]) {
  auto b = affs_from_wire(bases, n);
  auto s = frs_from_wire(scalars, n);
  jac_to_wire(g1_msm_naive(b.data(), s.data(), n), out);
}
// util.rs:25-29
void orc_g1_msm_jac(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[144]) {
  std::vector<G1> b(n);
  for (size_t i = 0; i < n; i++) b[i] = jac_from_wire(bases + 144 * i);
  jac_to_wire(g1_msm_from_projective(b, frs_from_wire(scalars, n)), out);
}
// inner_product_argument.rs:177-178 / same_multiscalar_argument.rs:128-130, in place on PL
void orc_g1_fold(uint8_t* PL, const uint8_t* PR, const uint8_t* gamma, size_t half) {
  Fr g = fr_from_wire(gamma);
  for (size_t i = 0; i < half; i++) {
    G1Aff l = aff_from_wire(PL + 96 * i), r = aff_from_wire(PR + 96 * i);
    aff_to_wire(g1_to_affine(g1_add(G1::from_affine(l), g1_mul(r, g))), PL + 96 * i);
  }
}
// grand_product_argument.rs:90-102 / util.rs:94-95: out[i] = affine(scalars[i] * P[i]);
// scalar_stride = 0 broadcasts one scalar
void orc_g1_scale(const uint8_t* P, const uint8_t* scalars, size_t scalar_stride, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; i++)
    aff_to_wire(g1_to_affine(g1_mul(aff_from_wire(P + 96 * i), fr_from_wire(scalars + scalar_stride * i))), out + 96 * i);
}
void orc_g1_to_affine(const uint8_t* jac, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; i++) aff_to_wire(g1_to_affine(jac_from_wire(jac + 144 * i)), out + 96 * i);
}
void orc_g1_add_jac(const uint8_t* a, const uint8_t* b, uint8_t* out) { jac_to_wire(g1_add(jac_from_wire(a), jac_from_wire(b)), out); }
int orc_g1_eq_jac(const uint8_t* a, const uint8_t* b) { return g1_eq(jac_from_wire(a), jac_from_wire(b)) ? 1 : 0; }
void orc_g1_compress(const uint8_t* aff, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; i++) g1_compress(aff_from_wire(aff + 96 * i), out + 48 * i);
}
void orc_g1_compress_jac(const uint8_t* jac, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; i++) g1_compress(jac_from_wire(jac + 144 * i), out + 48 * i);
}
void orc_set_strict_infinity(int strict) { g1_strict_infinity() = strict != 0; }
int orc_g1_decompress(const uint8_t* in, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; i++) {
    G1Aff p;
    if (!g1_decompress(in + 48 * i, &p)) return -1;
    aff_to_wire(p, out + 96 * i);
  }
  return 0;
}

// ---- CRS / instance generation following the reference's own recipes ----
// crs.rs:61-69: ell + 7 affine points from StdRng(0)
void orc_generate_crs_points(size_t ell, uint8_t* out) {
  StdRng rng(0);
  for (size_t i = 0; i < ell + N_BLINDERS + CRS_EXTRA_POINTS; i++) aff_to_wire(g1_to_affine(rand_g1(rng)), out + 96 * i);
}
// /root/reference/tests/crs.rs:13-52 `generate_random_points(num_points, seed)`: try-and-increment hash to the curve.
//   hash = SHA-256(seed || be64(increment)); x = the integer whose four low 64-bit limbs are the big-endian words of the hash
//   (limb i = hash[8i..8i+8], limbs 4 and 5 zero: `BigInteger384::new(x)`, `Fp384::new` converts it to Montgomery form);
//   `get_point_from_x_unchecked(x, false)`: the point with the lexicographically smaller y, if x^3 + 4 is a square;
//   `mul_by_cofactor()` (the full cofactor h, not h_eff), kept if in the subgroup (always; the identity would pass too).
// Returns the number of increments consumed (the reference's second tuple element).
uint64_t orc_hash_to_curve_points(const uint8_t* seed, size_t seed_len, size_t num_points, uint8_t* out) {
  static const uint64_t COFACTOR[2] = {0x8c00aaab0000aaabULL, 0x396c8c005555e156ULL};
  uint64_t increment = 0;
  size_t found = 0;
  std::vector<uint8_t> msg(seed, seed + seed_len);
  msg.resize(seed_len + 8);
  while (found != num_points) {
    for (int i = 0; i < 8; i++) msg[seed_len + i] = (uint8_t)(increment >> (56 - 8 * i));
    uint8_t h[32];
    sha256(msg.data(), msg.size(), h);
    uint64_t x[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 8; j++) x[i] = (x[i] << 8) | h[8 * i + j];
    G1Aff p;
    if (g1_point_from_x(Fp::from_canonical(x), false, &p)) {   // x < 2^256 < p
      const G1Aff z = g1_to_affine(g1_mul_limbs(G1::from_affine(p), COFACTOR, 2));
      if (g1_in_subgroup(z)) aff_to_wire(z, out + 96 * found++);
    }
    increment++;
  }
  return increment;
}
void orc_sha256(const uint8_t* msg, size_t len, uint8_t out[32]) { sha256(msg, len, out); }
// G_sum, H_sum of crs.rs:46-47 (96 B each)
void orc_crs_sums(size_t ell, const uint8_t* crs_points, uint8_t* g_sum, uint8_t* h_sum) {
  Crs crs;
  crs_from_wire(ell, crs_points, &crs);
  aff_to_wire(crs.G_sum, g_sum);
  aff_to_wire(crs.H_sum, h_sum);
}

// README.md:76-118 / curdleproofs.rs:337-377 recipe with `StdRng::seed_from_u64(seed)`:
// permutation.shuffle, k, vec_R, vec_S, shuffle_permute_and_commit_input, CurdleproofsProof::new, verify.
// Outputs (all caller-allocated): vec_R/S/T/U (ell*96), M (144), permutation (ell u32), k (32),
// vec_m_blinders (4*32), prover_rand ((3n+9)*32: the Fr draws `new` made, in order), proof bytes,
// verifier_rand (8*32). Returns the verifier's verdict (1/0) or <0 on error.
int orc_make_instance(size_t ell, uint64_t seed, const uint8_t* crs_points, uint8_t* vec_R, uint8_t* vec_S, uint8_t* vec_T, uint8_t* vec_U,
                      uint8_t* M, uint32_t* permutation, uint8_t* k_out, uint8_t* vec_m_blinders, uint8_t* prover_rand,
                      uint8_t* proof_out, size_t proof_cap, size_t* proof_len, uint8_t* verifier_rand) {
  Crs crs;
  if (!crs_from_wire(ell, crs_points, &crs)) return -1;
  StdRng rng(seed);
  std::vector<uint32_t> perm(ell);
  for (size_t i = 0; i < ell; i++) perm[i] = (uint32_t)i;
  shuffle(rng, perm);
  Fr k = rand_fr(rng);
  std::vector<G1Aff> R(ell), S(ell), T, U;
  for (auto& p : R) p = g1_to_affine(rand_g1(rng));
  for (auto& p : S) p = g1_to_affine(rand_g1(rng));
  G1 Mj;
  std::vector<Fr> mb;
  FrDraw plain = [&rng]() { return rand_fr(rng); };
  shuffle_permute_and_commit_input(crs, R, S, perm, k, plain, &T, &U, &Mj, &mb);

  Recorder prec{&rng, {}};
  CurdleproofsProof pf = curdleproofs_new(crs, R, S, T, U, Mj, perm, k, mb, std::ref(prec));
  std::vector<uint8_t> bytes = curdleproofs_serialize(pf);
  Recorder vrec{&rng, {}};
  bool ok = curdleproofs_verify(pf, crs, R, S, T, U, Mj, std::ref(vrec));

  size_t n = ell + N_BLINDERS;
  if (prec.log.size() != 3 * n + 9) return -3;
  if (ok && vrec.log.size() != 8) return -4;
  if (bytes.size() > proof_cap) return -2;
  for (size_t i = 0; i < ell; i++) {
    aff_to_wire(R[i], vec_R + 96 * i);
    aff_to_wire(S[i], vec_S + 96 * i);
    aff_to_wire(T[i], vec_T + 96 * i);
    aff_to_wire(U[i], vec_U + 96 * i);
  }
  jac_to_wire(Mj, M);
  memcpy(permutation, perm.data(), 4 * ell);
  fr_to_wire(k, k_out);
  for (size_t i = 0; i < N_BLINDERS; i++) fr_to_wire(mb[i], vec_m_blinders + 32 * i);
  for (size_t i = 0; i < prec.log.size(); i++) fr_to_wire(prec.log[i], prover_rand + 32 * i);
  memcpy(proof_out, bytes.data(), bytes.size());
  *proof_len = bytes.size();
  for (size_t i = 0; i < vrec.log.size() && i < 8; i++) fr_to_wire(vrec.log[i], verifier_rand + 32 * i);
  return ok ? 1 : 0;
}

// util.rs:83-106 with caller-supplied blinders (4 Fr)
int orc_shuffle_permute_and_commit_input(size_t ell, const uint8_t* crs_points, const uint8_t* vec_R, const uint8_t* vec_S,
                                         const uint32_t* permutation, const uint8_t* k, const uint8_t* m_blinders, uint8_t* vec_T,
                                         uint8_t* vec_U, uint8_t* M) {
  Crs crs;
  if (!crs_from_wire(ell, crs_points, &crs)) return -1;
  std::vector<G1Aff> T, U;
  G1 Mj;
  std::vector<Fr> mb;
  Replayer rp{m_blinders, N_BLINDERS};
  shuffle_permute_and_commit_input(crs, affs_from_wire(vec_R, ell), affs_from_wire(vec_S, ell),
                                   std::vector<uint32_t>(permutation, permutation + ell), fr_from_wire(k), std::ref(rp), &T, &U, &Mj, &mb);
  for (size_t i = 0; i < ell; i++) {
    aff_to_wire(T[i], vec_T + 96 * i);
    aff_to_wire(U[i], vec_U + 96 * i);
  }
  jac_to_wire(Mj, M);
  return 0;
}

// curdleproofs.rs:59-184 with the 3n+9 Fr draws supplied by the caller; writes `serialize` bytes
int orc_prove(size_t ell, const uint8_t* crs_points, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T,
              const uint8_t* vec_U, const uint8_t* M, const uint32_t* permutation, const uint8_t* k, const uint8_t* m_blinders,
              const uint8_t* rand, size_t n_rand, uint8_t* proof_out, size_t proof_cap, size_t* proof_len) {
  Crs crs;
  if (!crs_from_wire(ell, crs_points, &crs)) return -1;
  size_t n = ell + N_BLINDERS;
  if (n & (n - 1)) return -5;
  Replayer rp{rand, n_rand};
  CurdleproofsProof pf = curdleproofs_new(crs, affs_from_wire(vec_R, ell), affs_from_wire(vec_S, ell), affs_from_wire(vec_T, ell),
                                          affs_from_wire(vec_U, ell), jac_from_wire(M),
                                          std::vector<uint32_t>(permutation, permutation + ell), fr_from_wire(k),
                                          frs_from_wire(m_blinders, N_BLINDERS), std::ref(rp));
  if (rp.overrun) return -3;
  std::vector<uint8_t> bytes = curdleproofs_serialize(pf);
  if (bytes.size() > proof_cap) return -2;
  memcpy(proof_out, bytes.data(), bytes.size());
  *proof_len = bytes.size();
  return 0;
}

// curdleproofs.rs:197-298 on serialized proof bytes (deserialize with log2_n as whisk.rs:115-116 does).
// Returns 1 = Ok, 0 = VerificationError, -1 = deserialisation error
int orc_verify(size_t ell, const uint8_t* crs_points, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T,
               const uint8_t* vec_U, const uint8_t* M, const uint8_t* proof, size_t proof_len, const uint8_t* rand, size_t n_rand) {
  Crs crs;
  if (!crs_from_wire(ell, crs_points, &crs)) return -1;
  CurdleproofsProof pf;
  if (!curdleproofs_deserialize(proof, proof_len, crs.log2_n(), &pf)) return -1;
  Replayer rp{rand, n_rand};
  bool ok = curdleproofs_verify(pf, crs, affs_from_wire(vec_R, ell), affs_from_wire(vec_S, ell), affs_from_wire(vec_T, ell),
                                affs_from_wire(vec_U, ell), jac_from_wire(M), std::ref(rp));
  return ok ? 1 : 0;
}

// whisk.rs:228-263 / :183-226 with the caller's draws (tracker = r_G || k_r_G, 96 bytes); returns 1 / 0 / -1 (deserialisation)
int orc_whisk_tracker_proof(const uint8_t tracker[96], const uint8_t* k, const uint8_t* blinder, uint8_t out[128]) {
  WhiskTracker t;
  memcpy(t.r_G, tracker, 48);
  memcpy(t.k_r_G, tracker + 48, 48);
  Fr b = fr_from_wire(blinder);
  FrDraw draw = [&b]() { return b; };
  return generate_whisk_tracker_proof(draw, t, fr_from_wire(k), out) ? 1 : -1;
}
int orc_whisk_tracker_verify(const uint8_t tracker[96], const uint8_t k_commitment[48], const uint8_t proof[128]) {
  WhiskTracker t;
  memcpy(t.r_G, tracker, 48);
  memcpy(t.k_r_G, tracker + 48, 48);
  return is_valid_whisk_tracker_proof(t, k_commitment, proof);
}
// whisk.rs:106-130 with the verifier's draws replayed from `rand`
int orc_whisk_shuffle_verify(size_t ell, const uint8_t* crs_points, const uint8_t* pre, const uint8_t* post, const uint8_t* proof, size_t proof_len,
                             const uint8_t* rand, size_t n_rand) {
  Crs crs;
  if (!crs_from_wire(ell, crs_points, &crs)) return -1;
  std::vector<WhiskTracker> a(ell), b(ell);
  for (size_t i = 0; i < ell; i++) {
    memcpy(a[i].r_G, pre + 96 * i, 48);
    memcpy(a[i].k_r_G, pre + 96 * i + 48, 48);
    memcpy(b[i].r_G, post + 96 * i, 48);
    memcpy(b[i].k_r_G, post + 96 * i + 48, 48);
  }
  Replayer rp{rand, n_rand};
  return is_valid_whisk_shuffle_proof(std::ref(rp), crs, a, b, proof, proof_len);
}

// Timing helper for bench.py's cpu_baseline leg: `reps` x (prove + verify) on one instance,
// single thread; returns seconds for prove and verify separately.
int orc_time_prove_verify(size_t ell, const uint8_t* crs_points, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T,
                          const uint8_t* vec_U, const uint8_t* M, const uint32_t* permutation, const uint8_t* k,
                          const uint8_t* m_blinders, const uint8_t* rand, size_t n_rand, const uint8_t* vrand, int reps,
                          double* prove_s, double* verify_s) {
  Crs crs;
  if (!crs_from_wire(ell, crs_points, &crs)) return -1;
  auto R = affs_from_wire(vec_R, ell), S = affs_from_wire(vec_S, ell), T = affs_from_wire(vec_T, ell), U = affs_from_wire(vec_U, ell);
  G1 Mj = jac_from_wire(M);
  std::vector<uint32_t> perm(permutation, permutation + ell);
  Fr kk = fr_from_wire(k);
  auto mb = frs_from_wire(m_blinders, N_BLINDERS);
  double tp = 0, tv = 0;
  int okc = 0;
  for (int r = 0; r < reps; r++) {
    Replayer rp{rand, n_rand};
    auto t0 = std::chrono::steady_clock::now();
    CurdleproofsProof pf = curdleproofs_new(crs, R, S, T, U, Mj, perm, kk, mb, std::ref(rp));
    auto t1 = std::chrono::steady_clock::now();
    Replayer vp{vrand, 8};
    bool ok = curdleproofs_verify(pf, crs, R, S, T, U, Mj, std::ref(vp));
    auto t2 = std::chrono::steady_clock::now();
    tp += std::chrono::duration<double>(t1 - t0).count();
    tv += std::chrono::duration<double>(t2 - t1).count();
    okc += ok;
  }
  *prove_s = tp;
  *verify_s = tv;
  return okc == reps ? 1 : 0;
}

}  // extern "C"
