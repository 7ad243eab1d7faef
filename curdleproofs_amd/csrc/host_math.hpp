// Host-side scalar-field helpers and the Fiat-Shamir transcript of the engine — product code.
//
// The protocol's Fr bookkeeping (challenge algebra, vector folds, inner products) and the Merlin
// transcript serialise every round of the proof; in this round they run on the host, beside the
// device kernels, using the same 32-bit-limb Montgomery code the kernels use (mont32.hpp).
// Mirrors /root/reference/src/transcript.rs:14-60 (CurdleproofsTranscript over merlin 3.0.0) and the
// Fr helpers of /root/reference/src/util.rs:32-79.
#pragma once
#include <cstring>
#include <string>
#include <vector>
#include "mont32.hpp"
#include "strobe.hpp"

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
// a^-1 mod r by the binary extended Euclidean algorithm on 4 x u64 (canonical integers, 0 < a < r): ~510 halvings
// and ~255 subtractions of 256-bit integers, ~3x faster than Fermat's a^(r-2) (380 Montgomery products).
inline void inv_canonical(const uint64_t* a_in, uint64_t* out) {
  uint64_t u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u[i] = a_in[i];
    v[i] = P[i];
  }
  auto is_one = [](const uint64_t* w) { return w[0] == 1 && (w[1] | w[2] | w[3]) == 0; };
  auto is_zero = [](const uint64_t* w) { return (w[0] | w[1] | w[2] | w[3]) == 0; };
  auto halve = [](uint64_t* w, uint64_t top) {
    w[0] = (w[0] >> 1) | (w[1] << 63);
    w[1] = (w[1] >> 1) | (w[2] << 63);
    w[2] = (w[2] >> 1) | (w[3] << 63);
    w[3] = (w[3] >> 1) | (top << 63);
  };
  auto halve_mod = [&](uint64_t* x) {
    uint64_t carry = 0;
    if (x[0] & 1) {
      u128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (u128)x[i] + P[i];
        x[i] = (uint64_t)c;
        c >>= 64;
      }
      carry = (uint64_t)c;
    }
    halve(x, carry);
  };
  auto geq = [](const uint64_t* a, const uint64_t* b) {
    for (int i = 3; i >= 0; i--)
      if (a[i] != b[i]) return a[i] > b[i];
    return true;
  };
  auto sub = [](uint64_t* a, const uint64_t* b) {
    uint64_t bw = 0;
    for (int i = 0; i < 4; i++) {
      const u128 d = (u128)a[i] - b[i] - bw;
      a[i] = (uint64_t)d;
      bw = (uint64_t)(d >> 64) & 1;
    }
    return bw;
  };
  auto sub_mod = [&](uint64_t* a, const uint64_t* b) {
    if (sub(a, b)) {
      u128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + P[i];
        a[i] = (uint64_t)c;
        c >>= 64;
      }
    }
  };
  if (is_zero(u)) {
    for (int i = 0; i < 4; i++) out[i] = 0;
    return;
  }
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1)) {
      halve(u, 0);
      halve_mod(x1);
    }
    while (!(v[0] & 1)) {
      halve(v, 0);
      halve_mod(x2);
    }
    if (geq(u, v)) {
      sub(u, v);
      sub_mod(x1, x2);
    } else {
      sub(v, u);
      sub_mod(x2, x1);
    }
  }
  const uint64_t* r = is_one(u) ? x1 : x2;
  for (int i = 0; i < 4; i++) out[i] = r[i];
}
// R^3 mod r (R = 2^256): Montgomery-form inverse = inv_canonical(X) * R^3 / R
static constexpr uint64_t R3[4] = {0xc62c1807439b73afULL, 0x1b3e0d188cf06990ULL, 0x73d13c71c7b5f418ULL, 0x6e2a5bb9c8db33e9ULL};
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
  S inverse() const {   // binary Euclid on the Montgomery value X = a R: X^-1 = a^-1 R^-1, times R^3 -> a^-1 R; 0 -> 0
    uint64_t x[4], y[4];
    memcpy(x, f.v, 32);
    fr64::inv_canonical(x, y);
    S t, r3;
    memcpy(t.f.v, y, 32);
    memcpy(r3.f.v, fr64::R3, 32);
    return t * r3;
  }
  S inverse_fermat() const {   // a^(r-2); kept as the cross-check of inverse() in the tests
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

// A random factor of the accumulated check as the caller hands it over (wire form: Montgomery residue, 4 x u64 LE): it
// must be a reduced, non-zero element — a zero factor would drop its check from the accumulated sum.
inline bool is_valid_factor(const uint8_t wire[32]) {
  uint64_t w[4];
  memcpy(w, wire, 32);
  if ((w[0] | w[1] | w[2] | w[3]) == 0) return false;
  for (int i = 3; i >= 0; i--)
    if (w[i] != fr64::P[i]) return w[i] < fr64::P[i];
  return false;   // == r
}

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

// ---- Merlin transcript (STROBE-128 / Keccak core shared with the device: strobe.hpp) ----
using cpx::keccak_f1600;

class Transcript {   // merlin::Transcript specialised to the operations the protocol performs
 public:
  explicit Transcript(const char* label) { s_.init(label); }
  void append_message(const char* label, const uint8_t* m, size_t len) { s_.append_message(label, m, len); }
  void challenge_bytes(const char* label, uint8_t* out, size_t len) { s_.challenge_bytes(label, out, len); }
  // --- CurdleproofsTranscript (transcript.rs:28-60) ---
  void append_scalar(const char* label, const S& x) {
    uint8_t b[32];
    x.to_le_bytes(b);
    append_message(label, b, 32);
  }
  void append_point_bytes(const char* label, const uint8_t comp[48]) { append_message(label, comp, 48); }
  // Vec<G1Affine>: u64-LE length prefix then the compressed points, as one message
  void append_point_vec_bytes(const char* label, const uint8_t* comp, size_t n) {
    const uint64_t n64 = n;
    s_.append_begin(label, 8 + 48 * n);
    s_.absorb(reinterpret_cast<const uint8_t*>(&n64), 8);
    s_.absorb(comp, 48 * n);
  }
  void append_scalar_vec(const char* label, const SVec& v) {
    const uint64_t n64 = v.size();
    s_.append_begin(label, 8 + 32 * v.size());
    s_.absorb(reinterpret_cast<const uint8_t*>(&n64), 8);
    for (size_t i = 0; i < v.size(); i++) {
      uint8_t b[32];
      v[i].to_le_bytes(b);
      s_.absorb(b, 32);
    }
  }
  S get_and_append_challenge(const char* label) {
    uint32_t e[8];
    s_.challenge_scalar_canonical(label, e);
    uint8_t b[32];
    memcpy(b, e, 32);
    S x;
    S::from_le_bytes(b, &x);
    return x;
  }
  SVec get_and_append_challenges(const char* label, size_t n) {
    SVec v(n);
    for (auto& x : v) x = get_and_append_challenge(label);
    return v;
  }

 private:
  Strobe s_;
};

}  // namespace host
}  // namespace cpx
