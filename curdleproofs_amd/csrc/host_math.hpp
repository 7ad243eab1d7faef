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

// ---- host-speed Fr: the same Montgomery residues (R = 2^256) handled as 4 x u64 limbs with 128-bit products.
// The 8 x u32 little-endian limb array of cpx::Fr and 4 x u64 limbs are the same bytes on a little-endian host.
namespace fr64 {
typedef unsigned __int128 u128;
static constexpr uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static constexpr uint64_t INV = 0xfffffffeffffffffULL;   // -r^-1 mod 2^64
// Montgomery product, "no-carry" CIOS: the top limb of r is < 2^63 - 1, so the two carry words of a round never
// overflow when they are added (t3 = A + C) and the extra carry limb of textbook CIOS disappears.  18.6 ns against
// 25.5 ns for the generic form on the build host; the per-proof Fr algebra of the engine is ~30 k of these.
inline void mul(const uint32_t* a32, const uint32_t* b32, uint32_t* out32) {
  uint64_t a[4], b[4];
  memcpy(a, a32, 32);
  memcpy(b, b32, 32);
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  for (int i = 0; i < 4; i++) {
    u128 c = (u128)a[0] * b[i] + t0;
    const uint64_t lo = (uint64_t)c;
    uint64_t A = (uint64_t)(c >> 64);
    const uint64_t m = lo * INV;
    c = (u128)m * P[0] + lo;
    uint64_t C = (uint64_t)(c >> 64);
    c = (u128)a[1] * b[i] + t1 + A;
    A = (uint64_t)(c >> 64);
    c = (u128)m * P[1] + (uint64_t)c + C;
    t0 = (uint64_t)c;
    C = (uint64_t)(c >> 64);
    c = (u128)a[2] * b[i] + t2 + A;
    A = (uint64_t)(c >> 64);
    c = (u128)m * P[2] + (uint64_t)c + C;
    t1 = (uint64_t)c;
    C = (uint64_t)(c >> 64);
    c = (u128)a[3] * b[i] + t3 + A;
    A = (uint64_t)(c >> 64);
    c = (u128)m * P[3] + (uint64_t)c + C;
    t2 = (uint64_t)c;
    C = (uint64_t)(c >> 64);
    t3 = A + C;
  }
  // conditional subtraction
  const uint64_t t[4] = {t0, t1, t2, t3};
  uint64_t s[4];
  uint64_t bw = 0;
  for (int i = 0; i < 4; i++) {
    const u128 d = (u128)t[i] - P[i] - bw;
    s[i] = (uint64_t)d;
    bw = (uint64_t)(d >> 64) & 1;
  }
  memcpy(out32, bw ? t : s, 32);
}
}  // namespace fr64

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
  S operator*(const S& o) const {
    S r;
    fr64::mul(f.v, o.f.v, r.f.v);
    return r;
  }
  S operator-() const { return S(fe_neg(f)); }
  S& operator+=(const S& o) { return *this = *this + o; }
  S& operator*=(const S& o) { return *this = *this * o; }
  bool is_zero() const { return f.is_zero(); }
  bool operator==(const S& o) const { return f.eq(o.f); }
  S pow_limbs(const uint64_t* e, int nl) const {
    S r = one();
    bool started = false;
    for (int i = nl * 64 - 1; i >= 0; i--) {
      if (started) r = r * r;
      if ((e[i >> 6] >> (i & 63)) & 1) {
        r = started ? r * *this : *this;
        started = true;
      }
    }
    return r;
  }
  S inverse() const {   // Fermat, a^(r-2); 0 -> 0
    const uint64_t e[4] = {fr64::P[0] - 2, fr64::P[1], fr64::P[2], fr64::P[3]};
    return pow_limbs(e, 4);
  }
  S pow_u64(uint64_t e) const { return pow_limbs(&e, 1); }
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
  uint64_t a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a04 = a[4], a05 = a[5], a06 = a[6], a07 = a[7], a08 = a[8], a09 = a[9],
           a10 = a[10], a11 = a[11], a12 = a[12], a13 = a[13], a14 = a[14], a15 = a[15], a16 = a[16], a17 = a[17], a18 = a[18],
           a19 = a[19], a20 = a[20], a21 = a[21], a22 = a[22], a23 = a[23], a24 = a[24];
  for (int r = 0; r < 24; r++) {
    // theta
    const uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22,
                   c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23, c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
    const uint64_t d0 = c4 ^ rol64(c1, 1), d1 = c0 ^ rol64(c2, 1), d2 = c1 ^ rol64(c3, 1), d3 = c2 ^ rol64(c4, 1), d4 = c3 ^ rol64(c0, 1);
    a00 ^= d0; a05 ^= d0; a10 ^= d0; a15 ^= d0; a20 ^= d0;
    a01 ^= d1; a06 ^= d1; a11 ^= d1; a16 ^= d1; a21 ^= d1;
    a02 ^= d2; a07 ^= d2; a12 ^= d2; a17 ^= d2; a22 ^= d2;
    a03 ^= d3; a08 ^= d3; a13 ^= d3; a18 ^= d3; a23 ^= d3;
    a04 ^= d4; a09 ^= d4; a14 ^= d4; a19 ^= d4; a24 ^= d4;
    // rho + pi: B[y + 5*((2x+3y)%5)] = rot(A[x+5y], r[x][y])
    const uint64_t b00 = a00, b10 = rol64(a01, 1), b20 = rol64(a02, 62), b05 = rol64(a03, 28), b15 = rol64(a04, 27);
    const uint64_t b16 = rol64(a05, 36), b01 = rol64(a06, 44), b11 = rol64(a07, 6), b21 = rol64(a08, 55), b06 = rol64(a09, 20);
    const uint64_t b07 = rol64(a10, 3), b17 = rol64(a11, 10), b02 = rol64(a12, 43), b12 = rol64(a13, 25), b22 = rol64(a14, 39);
    const uint64_t b23 = rol64(a15, 41), b08 = rol64(a16, 45), b18 = rol64(a17, 15), b03 = rol64(a18, 21), b13 = rol64(a19, 8);
    const uint64_t b14 = rol64(a20, 18), b24 = rol64(a21, 2), b09 = rol64(a22, 61), b19 = rol64(a23, 56), b04 = rol64(a24, 14);
    // chi
    a00 = b00 ^ (~b01 & b02); a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
    a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
    a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
    a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
    a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
    a00 ^= RC[r];
  }
  a[0] = a00; a[1] = a01; a[2] = a02; a[3] = a03; a[4] = a04; a[5] = a05; a[6] = a06; a[7] = a07; a[8] = a08; a[9] = a09;
  a[10] = a10; a[11] = a11; a[12] = a12; a[13] = a13; a[14] = a14; a[15] = a15; a[16] = a16; a[17] = a17; a[18] = a18; a[19] = a19;
  a[20] = a20; a[21] = a21; a[22] = a22; a[23] = a23; a[24] = a24;
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
