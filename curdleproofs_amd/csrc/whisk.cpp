// Whisk byte-level API (/root/reference/src/whisk.rs) on top of the engine — product code.
//
// Host glue only: every group operation below is a call into the device engine (decompression + subgroup check,
// scalar multiplications, MSMs, the batch prover / verifier, normalisation + compression); the host hashes the
// six-point transcript of the tracker proofs and moves bytes.  Randomness stays with the caller (SURVEY 8b RNG
// contract): every `Fr::rand(rng)` / `shuffle(rng)` the reference performs is an argument, in the reference's order.
#include <cstring>
#include <vector>
#include "engine.hpp"

namespace cpx {

using host::S;
using host::Transcript;

namespace {
// ark_bls12_381::g1::G1Affine::generator(), compressed (whisk.rs:363-368 pins these bytes)
const uint8_t GEN_COMP[48] = {0x97, 0xf1, 0xd3, 0xa7, 0x31, 0x97, 0xd7, 0x94, 0x26, 0x95, 0x63, 0x8c, 0x4f, 0xa9, 0xac, 0x0f,
                              0xc3, 0x68, 0x8c, 0x4f, 0x97, 0x74, 0xb9, 0x05, 0xa1, 0x4e, 0x3a, 0x3f, 0x17, 0x1b, 0xac, 0x58,
                              0x6c, 0x55, 0xe8, 0x3f, 0xf9, 0x7a, 0x1a, 0xef, 0xfb, 0x3a, 0xf0, 0x0a, 0xdb, 0x22, 0xc6, 0xbb};

struct DeserializeError {};   // ark_serialize::SerializationError

void jac_from_aff(const Aff* a, size_t n, std::vector<Jac>& out) {
  out.resize(n);
  for (size_t i = 0; i < n; i++) out[i] = Jac::from_affine(a[i]);
}
}  // namespace

const Aff& Engine::generator() {
  if (!have_gen_) {
    if (decompress(GEN_COMP, 1, reinterpret_cast<uint8_t*>(&gen_), 1) != CPX_OK) throw std::logic_error("generator decoding");
    have_gen_ = true;
  }
  return gen_;
}

// affine points -> 48-byte compressed encodings (G1Affine::serialize_compressed)
void Engine::compress_affine(const Aff* pts, size_t n, uint8_t* out) {
  std::vector<Jac> j;
  jac_from_aff(pts, n, j);
  normalize(reinterpret_cast<const uint8_t*>(j.data()), n, nullptr, out);
}

// whisk.rs:265-277 unzip_trackers: r_G (48 B) || k_r_G (48 B) per tracker -> two affine vectors; false on a bad encoding
bool Engine::unzip_trackers(const uint8_t* trackers, size_t n, std::vector<Aff>& vec_r, std::vector<Aff>& vec_s) {
  std::vector<Aff> both(2 * n);
  if (decompress(trackers, 2 * n, reinterpret_cast<uint8_t*>(both.data()), 1) != CPX_OK) return false;
  vec_r.resize(n);
  vec_s.resize(n);
  for (size_t i = 0; i < n; i++) {
    vec_r[i] = both[2 * i];
    vec_s[i] = both[2 * i + 1];
  }
  return true;
}

// whisk.rs:144-179 generate_whisk_shuffle_proof (+ util.rs:83-106 shuffle_permute_and_commit_input)
int Engine::whisk_generate_shuffle_proof(const uint8_t* pre_trackers, const uint32_t* permutation, const uint8_t* k, const uint8_t* vec_m_blinders,
                                         const uint8_t* rand, uint8_t* post_trackers_out, uint8_t* proof_out) {
  if (!ell_) throw std::logic_error("set_crs first");
  const size_t ell = ell_, n = n_;
  std::vector<Aff> vec_r, vec_s;
  if (!unzip_trackers(pre_trackers, ell, vec_r, vec_s)) return CPX_ERR_DESERIALIZE;
  // vec_T = permute(k * vec_R), vec_U = permute(k * vec_S)   (util.rs:94-97)
  std::vector<Aff> kr(ell), ks(ell), vec_t(ell), vec_u(ell);
  scale(reinterpret_cast<const uint8_t*>(vec_r.data()), k, 0, ell, reinterpret_cast<uint8_t*>(kr.data()));
  scale(reinterpret_cast<const uint8_t*>(vec_s.data()), k, 0, ell, reinterpret_cast<uint8_t*>(ks.data()));
  for (size_t i = 0; i < ell; i++) {
    vec_t[i] = kr[permutation[i]];
    vec_u[i] = ks[permutation[i]];
  }
  // M = msm(vec_G, sigma) + msm(vec_H, blinders)   (util.rs:99-104)
  std::vector<Fr> sc(n);
  for (size_t i = 0; i < ell; i++) sc[i] = S::from_u64(permutation[i]).f;
  memcpy(&sc[ell], vec_m_blinders, 4 * sizeof(Fr));
  Jac M;
  msm(reinterpret_cast<const uint8_t*>(crs_host_.data()), reinterpret_cast<const uint8_t*>(sc.data()), n, reinterpret_cast<uint8_t*>(&M));
  // CurdleproofsProof::new
  batch_load(1, reinterpret_cast<const uint8_t*>(vec_r.data()), reinterpret_cast<const uint8_t*>(vec_s.data()), reinterpret_cast<const uint8_t*>(vec_t.data()),
             reinterpret_cast<const uint8_t*>(vec_u.data()), reinterpret_cast<const uint8_t*>(&M));
  batch_prove(permutation, k, vec_m_blinders, rand, proof_out + 48);
  normalize(reinterpret_cast<const uint8_t*>(&M), 1, nullptr, proof_out);   // WhiskShuffleProof::serialize: M first (whisk.rs:87-91)
  // zip_trackers (whisk.rs:279-293)
  std::vector<Aff> zipped(2 * ell);
  for (size_t i = 0; i < ell; i++) {
    zipped[2 * i] = vec_t[i];
    zipped[2 * i + 1] = vec_u[i];
  }
  compress_affine(zipped.data(), 2 * ell, post_trackers_out);
  return CPX_OK;
}

// whisk.rs:106-130 is_valid_whisk_shuffle_proof
int Engine::whisk_is_valid_shuffle_proof(const uint8_t* pre_trackers, const uint8_t* post_trackers, const uint8_t* proof, const uint8_t* rand, int* valid) {
  if (!ell_) throw std::logic_error("set_crs first");
  const size_t ell = ell_;
  *valid = 0;
  std::vector<Aff> vec_r, vec_s, vec_t, vec_u;
  if (!unzip_trackers(pre_trackers, ell, vec_r, vec_s) || !unzip_trackers(post_trackers, ell, vec_t, vec_u)) return CPX_ERR_DESERIALIZE;
  Aff m_aff;
  if (decompress(proof, 1, reinterpret_cast<uint8_t*>(&m_aff), 1) != CPX_OK) return CPX_ERR_DESERIALIZE;   // G1Projective::deserialize_compressed
  const Jac M = Jac::from_affine(m_aff);
  batch_load(1, reinterpret_cast<const uint8_t*>(vec_r.data()), reinterpret_cast<const uint8_t*>(vec_s.data()), reinterpret_cast<const uint8_t*>(vec_t.data()),
             reinterpret_cast<const uint8_t*>(vec_u.data()), reinterpret_cast<const uint8_t*>(&M));
  int verdict = CPX_ERR_INTERNAL;
  batch_verify(proof + 48, rand, &verdict);
  if (verdict == CPX_ERR_DESERIALIZE) return CPX_ERR_DESERIALIZE;   // CurdleproofsProof::deserialize failed: Err(SerializationError)
  *valid = verdict == CPX_OK ? 1 : 0;                               // .verify(...).is_ok()
  return CPX_OK;
}

namespace {
// the six-point transcript of both tracker-proof functions (whisk.rs:204-218, :243-257)
S tracker_challenge(const uint8_t comp6[6 * 48]) {
  Transcript tr("whisk_opening_proof");
  for (int i = 0; i < 6; i++) tr.append_point_bytes("tracker_opening_proof", comp6 + 48 * i);
  return tr.get_and_append_challenge("tracker_opening_proof_challenge");
}
}  // namespace

// whisk.rs:228-263 generate_whisk_tracker_proof; `blinder` is the function's one Fr::rand draw
int Engine::whisk_generate_tracker_proof(const uint8_t tracker[96], const uint8_t k[32], const uint8_t blinder[32], uint8_t proof_out[128]) {
  Aff tr[2];   // r_G, k_r_G
  if (decompress(tracker, 2, reinterpret_cast<uint8_t*>(tr), 1) != CPX_OK) return CPX_ERR_DESERIALIZE;
  const Aff& G = generator();
  // k_G = k G, A = blinder G, B = blinder r_G
  const Aff bases[3] = {G, G, tr[0]};
  uint8_t scal[3 * 32];
  memcpy(scal, k, 32);
  memcpy(scal + 32, blinder, 32);
  memcpy(scal + 64, blinder, 32);
  Aff out[3];
  scale(reinterpret_cast<const uint8_t*>(bases), scal, 32, 3, reinterpret_cast<uint8_t*>(out));
  const Aff six[6] = {out[0], G, tr[1], tr[0], out[1], out[2]};
  uint8_t comp[6 * 48];
  compress_affine(six, 6, comp);
  const S challenge = tracker_challenge(comp);
  S kk, bl;
  memcpy(kk.f.v, k, 32);
  memcpy(bl.f.v, blinder, 32);
  const S s = bl - challenge * kk;
  memcpy(proof_out, comp + 4 * 48, 96);   // A, B (TrackerProof::serialize_compressed, whisk.rs:69-73)
  s.to_le_bytes(proof_out + 96);
  return CPX_OK;
}

// whisk.rs:183-226 is_valid_whisk_tracker_proof
int Engine::whisk_is_valid_tracker_proof(const uint8_t tracker[96], const uint8_t k_commitment[48], const uint8_t proof[128], int* valid) {
  *valid = 0;
  S s;
  if (!S::from_le_bytes(proof + 96, &s)) return CPX_ERR_DESERIALIZE;   // TrackerProof::deserialize_compressed
  uint8_t comp5[5 * 48];
  memcpy(comp5, proof, 96);              // A, B
  memcpy(comp5 + 96, tracker + 48, 48);  // k_r_G
  memcpy(comp5 + 144, tracker, 48);      // r_G
  memcpy(comp5 + 192, k_commitment, 48); // k_G
  Aff pts[5];
  if (decompress(comp5, 5, reinterpret_cast<uint8_t*>(pts), 1) != CPX_OK) return CPX_ERR_DESERIALIZE;
  const Aff &A = pts[0], &B = pts[1], &k_r_G = pts[2], &r_G = pts[3], &k_G = pts[4];
  const Aff& G = generator();
  const Aff six[6] = {k_G, G, k_r_G, r_G, A, B};
  uint8_t comp[6 * 48];
  compress_affine(six, 6, comp);   // the canonical encodings the reference hashes (serialize_compressed of the decoded points)
  const S challenge = tracker_challenge(comp);
  // A' = s G + c k_G,  B' = s r_G + c k_r_G
  const Aff ba[2] = {G, k_G}, bb[2] = {r_G, k_r_G};
  Fr sc[2] = {s.f, challenge.f};
  Jac res[2];
  msm(reinterpret_cast<const uint8_t*>(ba), reinterpret_cast<const uint8_t*>(sc), 2, reinterpret_cast<uint8_t*>(&res[0]));
  msm(reinterpret_cast<const uint8_t*>(bb), reinterpret_cast<const uint8_t*>(sc), 2, reinterpret_cast<uint8_t*>(&res[1]));
  uint8_t got[2 * 48];
  normalize(reinterpret_cast<const uint8_t*>(res), 2, nullptr, got);
  *valid = (memcmp(got, comp + 4 * 48, 96) == 0) ? 1 : 0;
  return CPX_OK;
}

}  // namespace cpx
