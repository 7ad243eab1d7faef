// Host-side scalar-field helpers and the Fiat-Shamir transcript of the engine — product code.
//
// The protocol's Fr bookkeeping (challenge algebra, vector folds, inner products) and the Merlin
// transcript serialise every round of the proof; in this round they run on the host, beside the
// device kernels, using the same 32-bit-limb Montgomery code the kernels use (mont32.cuh).
// Mirrors /root/reference/src/transcript.rs:14-60 (CurdleproofsTranscript over merlin 3.0.0) and the
// Fr helpers of /root/reference/src/util.rs:32-79.
#pragma once
#include <cstring>
#include <string>
#include <vector>
#include "mont32.cuh"

namespace cpx {
namespace host {

// ---- Fr sugar ----
struct S {   // a scalar in Montgomery form
  Fr f;
  S() : f(Fr::zero()) {}
  explicit S(const Fr& x) : f(x) {}
  static S zero() { return S(Fr::zero()); }
  static S one() { return S(Fr::one()); }
  static S from_u64(uint64_t x) {
    Fr c = Fr::zero();
    c.v[0] = (uint32_t)x;
    c.v[1] = (uint32_t)(x >> 32);
    return S(fe_to_mont(c));
  }
  S operator+(const S& o) const { return S(fe_add(f, o.f)); }
  S operator-(const S& o) const { return S(fe_sub(f, o.f)); }
  S operator*(const S& o) const { return S(fe_mul(f, o.f)); }
  S operator-() const { return S(fe_neg(f)); }
  S& operator+=(const S& o) { return *this = *this + o; }
  S& operator*=(const S& o) { return *this = *this * o; }
  bool is_zero() const { return f.is_zero(); }
  bool operator==(const S& o) const { return f.eq(o.f); }
  S inverse() const { return S(fe_inv(f)); }
  S pow_u64(uint64_t e) const {
    uint32_t w[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
    return S(fe_pow(f, w, 2));
  }
  void to_le_bytes(uint8_t out[32]) const {   // canonical little-endian (ark-serialize)
    Fr c = fe_from_mont(f);
    memcpy(out, c.v, 32);
  }
  // canonical little-endian -> S; false if >= r
  static bool from_le_bytes(const uint8_t in[32], S* out) {
    Fr c, m;
    memcpy(c.v, in, 32);
    for (int i = 0; i < 8; i++) m.v[i] = FrCfg::P[i];
    if (!fe_raw_gt(m, c)) return false;
    out->f = fe_to_mont(c);
    return true;
  }
};
typedef std::vector<S> SVec;

inline S inner_product(const S* a, const S* b, size_t n) {   // util.rs:66-73
  S c = S::zero();
  for (size_t i = 0; i < n; i++) c += a[i] * b[i];
  return c;
}
// Montgomery's trick; zeros stay zero (ark_ff::batch_inversion semantics)
inline void batch_inverse(SVec& v) {
  size_t n = v.size();
  SVec pre(n);
  S acc = S::one();
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    if (!v[i].is_zero()) acc *= v[i];
  }
  S inv = acc.inverse();
  for (size_t i = n; i-- > 0;) {
    if (v[i].is_zero()) continue;
    S t = inv * pre[i];
    inv *= v[i];
    v[i] = t;
  }
}

// ---- Keccak-f[1600] / STROBE-128 / Merlin ----
inline uint64_t rol64(uint64_t x, unsigned s) { return (x << s) | (x >> ((64 - s) & 63)); }

inline void keccak_f1600(uint64_t* a) {
  static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                  0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                  0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                  0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                  0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                  0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  // rho offsets indexed [x + 5y]
  static const unsigned RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  for (int r = 0; r < 24; r++) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
    // rho + pi: B[y, 2x+3y] = rot(A[x,y])
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = RHO[x + 5 * y] ? rol64(a[x + 5 * y], RHO[x + 5 * y]) : a[x + 5 * y];
    for (int y = 0; y < 5; y++)
      for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= RC[r];
  }
}

class Transcript {   // merlin::Transcript specialised to the operations the protocol performs
 public:
  explicit Transcript(const char* label) {
    memset(st_, 0, sizeof st_);
    static const uint8_t init[18] = {1, 168, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    memcpy(st_, init, 18);
    permute();
    pos_ = pos_begin_ = 0;
    meta_ad("Merlin v1.0", 11, false);
    append_message("dom-sep", (const uint8_t*)label, strlen(label));
  }
  void append_message(const char* label, const uint8_t* m, size_t len) {
    uint8_t l4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    meta_ad(label, strlen(label), false);
    meta_ad(l4, 4, true);
    begin_op(FLAG_A, false);
    absorb(m, len);
  }
  void challenge_bytes(const char* label, uint8_t* out, size_t len) {
    uint8_t l4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    meta_ad(label, strlen(label), false);
    meta_ad(l4, 4, true);
    begin_op(FLAG_I | FLAG_A | FLAG_C, false);
    for (size_t i = 0; i < len; i++) {
      out[i] = st_[pos_];
      st_[pos_] = 0;
      if (++pos_ == RATE) run_f();
    }
  }
  // --- CurdleproofsTranscript (transcript.rs:28-60) ---
  void append_scalar(const char* label, const S& x) {
    uint8_t b[32];
    x.to_le_bytes(b);
    append_message(label, b, 32);
  }
  void append_point_bytes(const char* label, const uint8_t comp[48]) { append_message(label, comp, 48); }
  // Vec<G1Affine>: u64-LE length prefix then the compressed points, as one message
  void append_point_vec_bytes(const char* label, const uint8_t* comp, size_t n) {
    std::vector<uint8_t> b(8 + 48 * n);
    uint64_t n64 = n;
    memcpy(b.data(), &n64, 8);
    memcpy(b.data() + 8, comp, 48 * n);
    append_message(label, b.data(), b.size());
  }
  void append_scalar_vec(const char* label, const SVec& v) {
    std::vector<uint8_t> b(8 + 32 * v.size());
    uint64_t n64 = v.size();
    memcpy(b.data(), &n64, 8);
    for (size_t i = 0; i < v.size(); i++) v[i].to_le_bytes(&b[8 + 32 * i]);
    append_message(label, b.data(), b.size());
  }
  S get_and_append_challenge(const char* label) {
    for (;;) {
      uint8_t buf[64];
      challenge_bytes(label, buf, 64);
      buf[31] &= 0x7f;   // Fr::from_random_bytes keeps 255 bits of the first 32 bytes
      S e;
      if (S::from_le_bytes(buf, &e) && !e.is_zero()) {
        append_scalar(label, e);
        return e;
      }
    }
  }
  SVec get_and_append_challenges(const char* label, size_t n) {
    SVec v(n);
    for (auto& x : v) x = get_and_append_challenge(label);
    return v;
  }

 private:
  static constexpr uint8_t RATE = 166, FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_M = 16, FLAG_K = 32;
  alignas(8) uint8_t st_[200];
  uint8_t pos_, pos_begin_;

  void permute() { keccak_f1600(reinterpret_cast<uint64_t*>(st_)); }
  void run_f() {
    st_[pos_] ^= pos_begin_;
    st_[pos_ + 1] ^= 0x04;
    st_[RATE + 1] ^= 0x80;
    permute();
    pos_ = pos_begin_ = 0;
  }
  void absorb(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
      st_[pos_] ^= d[i];
      if (++pos_ == RATE) run_f();
    }
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;
    uint8_t hdr[2] = {pos_begin_, flags};
    pos_begin_ = pos_ + 1;
    absorb(hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && pos_ != 0) run_f();
  }
  void meta_ad(const void* d, size_t n, bool more) {
    begin_op(FLAG_M | FLAG_A, more);
    absorb(static_cast<const uint8_t*>(d), n);
  }
};

}  // namespace host
}  // namespace cpx
