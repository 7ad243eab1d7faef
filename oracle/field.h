// ORACLE (test infrastructure — never linked into the product path).
//
// CPU restatement of the prime-field arithmetic the reference gets from
// ark-ff ^0.4 (Cargo.toml:23) through ark-bls12-381 ^0.4 (Cargo.toml:24):
//   Fp  = BLS12-381 base field, 6 x u64 Montgomery limbs, R = 2^384
//   Fr  = BLS12-381 scalar field, 4 x u64 Montgomery limbs, R = 2^256
// Reference call sites: every `Fr`/`G1*` use in /root/reference/src/*.rs
// (e.g. util.rs:66-73 inner_product, grand_product_argument.rs:86 inverse,
// inner_product_argument.rs:234 batch_inversion).
//
// The reference's dependency is not vendored under /root/reference; this file
// restates the published algorithm (CIOS Montgomery multiplication; Fermat
// inversion; p = 3 mod 4 square root) and is pinned through the reference's
// known-answer tests (whisk.rs:355-456) in tests/test_oracle_kat.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace orc {

typedef unsigned __int128 u128;

template <int N> static inline int big_cmp(const uint64_t* a, const uint64_t* b) {
  for (int i = N - 1; i >= 0; i--) {
    if (a[i] < b[i]) return -1;
    if (a[i] > b[i]) return 1;
  }
  return 0;
}
template <int N> static inline uint64_t big_add(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  u128 c = 0;
  for (int i = 0; i < N; i++) {
    c += (u128)a[i] + b[i];
    r[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
template <int N> static inline uint64_t big_sub(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t borrow = 0;
  for (int i = 0; i < N; i++) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
template <int N> static inline bool big_is_zero(const uint64_t* a) {
  uint64_t o = 0;
  for (int i = 0; i < N; i++) o |= a[i];
  return o == 0;
}

// Montgomery field over an N-limb modulus. P supplies MOD (little-endian limbs).
template <class P> struct Mont {
  typedef P Params;
  static constexpr int N = P::N;
  uint64_t v[N];

  // ---- derived constants (computed once) ----
  struct Consts {
    uint64_t inv;        // -MOD^{-1} mod 2^64
    uint64_t one[N];     // R mod MOD
    uint64_t r2[N];      // R^2 mod MOD
    Consts() {
      uint64_t x = 1;    // Newton: x <- x(2 - m x)
      for (int i = 0; i < 6; i++) x *= 2 - P::MOD[0] * x;
      inv = (uint64_t)0 - x;
      uint64_t t[N];
      memset(t, 0, sizeof t);
      t[0] = 1;
      for (int i = 0; i < 2 * 64 * N; i++) {   // t = 2^i mod MOD
        uint64_t carry = big_add<N>(t, t, t);
        if (carry || big_cmp<N>(t, P::MOD) >= 0) big_sub<N>(t, t, P::MOD);
        if (i == 64 * N - 1) memcpy(one, t, sizeof t);
      }
      memcpy(r2, t, sizeof t);
    }
  };
  static const Consts& C() {
    static const Consts c;
    return c;
  }

  static Mont zero() {
    Mont r;
    memset(r.v, 0, sizeof r.v);
    return r;
  }
  static Mont one() {
    Mont r;
    memcpy(r.v, C().one, sizeof r.v);
    return r;
  }
  bool is_zero() const { return big_is_zero<N>(v); }
  bool operator==(const Mont& o) const { return memcmp(v, o.v, sizeof v) == 0; }
  bool operator!=(const Mont& o) const { return !(*this == o); }

  Mont operator+(const Mont& o) const {
    Mont r;
    uint64_t c = big_add<N>(r.v, v, o.v);
    if (c || big_cmp<N>(r.v, P::MOD) >= 0) big_sub<N>(r.v, r.v, P::MOD);
    return r;
  }
  Mont operator-(const Mont& o) const {
    Mont r;
    if (big_sub<N>(r.v, v, o.v)) big_add<N>(r.v, r.v, P::MOD);
    return r;
  }
  Mont operator-() const {
    if (is_zero()) return *this;
    Mont r;
    big_sub<N>(r.v, P::MOD, v);
    return r;
  }
  Mont dbl() const { return *this + *this; }

  // CIOS Montgomery product: a*b*R^{-1} mod MOD
  static Mont mont_mul(const uint64_t* a, const uint64_t* b) {
    const uint64_t inv = C().inv;
    uint64_t t[N + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < N; i++) {
      u128 c = 0;
      for (int j = 0; j < N; j++) {
        c += (u128)a[j] * b[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[N];
      t[N] = (uint64_t)c;
      t[N + 1] = (uint64_t)(c >> 64);
      uint64_t m = t[0] * inv;
      c = (u128)m * P::MOD[0] + t[0];
      c >>= 64;
      for (int j = 1; j < N; j++) {
        c += (u128)m * P::MOD[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[N];
      t[N - 1] = (uint64_t)c;
      t[N] = t[N + 1] + (uint64_t)(c >> 64);
    }
    Mont r;
    if (t[N] || big_cmp<N>(t, P::MOD) >= 0) big_sub<N>(r.v, t, P::MOD);
    else memcpy(r.v, t, sizeof r.v);
    return r;
  }
  Mont operator*(const Mont& o) const { return mont_mul(v, o.v); }
  Mont sqr() const { return mont_mul(v, v); }
  Mont& operator+=(const Mont& o) { return *this = *this + o; }
  Mont& operator-=(const Mont& o) { return *this = *this - o; }
  Mont& operator*=(const Mont& o) { return *this = *this * o; }

  // canonical (non-Montgomery) integer <-> field element
  static Mont from_canonical(const uint64_t* limbs) {   // caller guarantees limbs < MOD
    return mont_mul(limbs, C().r2);
  }
  void to_canonical(uint64_t* out) const {
    uint64_t o[N];
    memset(o, 0, sizeof o);
    o[0] = 1;
    Mont r = mont_mul(v, o);
    memcpy(out, r.v, sizeof r.v);
  }
  static Mont from_u64(uint64_t x) {
    uint64_t l[N];
    memset(l, 0, sizeof l);
    l[0] = x;
    return from_canonical(l);
  }
  // ark-ff `Fp::cmp` compares canonical integers (used for the "lexicographically
  // largest y" flag, ark-bls12-381 g1 serialization; whisk.rs:363-368 pins it).
  int cmp_canonical(const Mont& o) const {
    uint64_t a[N], b[N];
    to_canonical(a);
    o.to_canonical(b);
    return big_cmp<N>(a, b);
  }

  // exponent given as little-endian limbs
  Mont pow(const uint64_t* e, int nlimbs) const {
    Mont r = one();
    bool started = false;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
      if (started) r = r.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) {
        r = started ? r * *this : *this;
        started = true;
      }
    }
    return r;
  }
  Mont pow_u64(uint64_t e) const { return pow(&e, 1); }
  // Fermat inverse; zero maps to zero (callers check).
  Mont inverse() const {
    uint64_t e[N], two[N];
    memset(two, 0, sizeof two);
    two[0] = 2;
    big_sub<N>(e, P::MOD, two);
    return pow(e, N);
  }
};

struct FpParams {
  static constexpr int N = 6;
  static constexpr uint64_t MOD[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                      0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
};
struct FrParams {
  static constexpr int N = 4;
  static constexpr uint64_t MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                      0x73eda753299d7d48ULL};
};
typedef Mont<FpParams> Fp;
typedef Mont<FrParams> Fr;

// sqrt in Fp for p = 3 mod 4 (ark-ff SqrtPrecomputation::Case3Mod4): a^((p+1)/4), checked.
static inline bool fp_sqrt(const Fp& a, Fp* out) {
  uint64_t e[6], one[6] = {1, 0, 0, 0, 0, 0};
  big_add<6>(e, FpParams::MOD, one);   // p+1 (no overflow: p < 2^381)
  for (int i = 0; i < 6; i++) e[i] = (e[i] >> 2) | (i < 5 ? e[i + 1] << 62 : 0);
  Fp r = a.pow(e, 6);
  if (r.sqr() != a) return false;
  *out = r;
  return true;
}

// Fr <-> 32-byte little-endian canonical (ark-serialize; whisk.rs:355-361 pins it)
static inline void fr_to_bytes(const Fr& a, uint8_t out[32]) {
  uint64_t c[4];
  a.to_canonical(c);
  memcpy(out, c, 32);   // host is little-endian
}
// returns false if the integer is >= r
static inline bool fr_from_bytes(const uint8_t in[32], Fr* out) {
  uint64_t c[4];
  memcpy(c, in, 32);
  if (big_cmp<4>(c, FrParams::MOD) >= 0) return false;
  *out = Fr::from_canonical(c);
  return true;
}
// ark-ff `from_le_bytes_mod_order` for inputs of at most 32 bytes... general length (whisk.rs:344-346)
static inline Fr fr_from_le_bytes_mod_order(const uint8_t* in, size_t len) {
  // Horner over bytes, most significant first: acc = acc*256 + byte
  Fr acc = Fr::zero();
  Fr b256 = Fr::from_u64(256);
  for (size_t i = len; i-- > 0;) acc = acc * b256 + Fr::from_u64(in[i]);
  return acc;
}

}  // namespace orc
