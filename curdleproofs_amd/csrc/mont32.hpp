// Montgomery prime-field arithmetic on 32-bit limbs for gfx950 (CDNA4) — product code.
//
// BLS12-381 Fp (12 limbs, R = 2^384) and Fr (8 limbs, R = 2^256).  The Montgomery radix equals
// the one arkworks uses on 64-bit limbs, so a field element has the same bytes in memory on both
// sides of the C-ABI: no conversion at the boundary.
//
// Everything here is `__host__ __device__` and branch-light so that
//   * the HIP kernels (msm / fold / scale / normalise) keep a whole element in VGPRs
//     (12 registers per Fp, fully unrolled limb loops -> static register indexing), and
//   * the host engine uses the very same code for the Fr bookkeeping of the protocol, and
//   * tests can run the device math bit-for-bit on the CPU (tests/host_emul) against the oracle.
// The 32x32->64 multiply-accumulate `(u64)a*b + c` lowers to one v_mad_u64_u32 on gfx950.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CPX_HD __host__ __device__ __forceinline__
// Real (non-inlined) functions for the big bodies: one copy of the ~1.3k-instruction Montgomery
// product and of each point formula per code object keeps the kernels inside the instruction
// cache (a fully inlined mixed add is > 100 KB of code) and cuts compile time by an order of magnitude.
#define CPX_HD_FN __host__ __device__ __noinline__ inline
#define CPX_UNROLL _Pragma("unroll")
#else
#define CPX_HD inline __attribute__((always_inline))
#define CPX_HD_FN inline __attribute__((noinline))
#define CPX_UNROLL _Pragma("GCC unroll 16")
#endif

namespace cpx {

// ---- field configurations (little-endian 32-bit limbs) ----
struct FpCfg {
  static constexpr int N = 12;
  static constexpr uint32_t P[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                     0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
  // R mod p
  static constexpr uint32_t ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                       0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
  // R^2 mod p
  static constexpr uint32_t R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                      0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
  static constexpr uint32_t INV = 0xfffcfffdu;   // -p^{-1} mod 2^32
};
struct FrCfg {
  static constexpr int N = 8;
  static constexpr uint32_t P[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                    0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  static constexpr uint32_t ONE[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                      0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
  static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                     0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
  static constexpr uint32_t INV = 0xffffffffu;   // -r^{-1} mod 2^32
};

template <class C> struct Fe {
  static constexpr int N = C::N;
  typedef C Cfg;
  uint32_t v[N];

  static CPX_HD Fe zero() {
    Fe r;
    CPX_UNROLL for (int i = 0; i < N; i++) r.v[i] = 0;
    return r;
  }
  static CPX_HD Fe one() {
    Fe r;
    CPX_UNROLL for (int i = 0; i < N; i++) r.v[i] = C::ONE[i];
    return r;
  }
  static CPX_HD Fe r2() {
    Fe r;
    CPX_UNROLL for (int i = 0; i < N; i++) r.v[i] = C::R2[i];
    return r;
  }
  CPX_HD bool is_zero() const {
    uint32_t o = 0;
    CPX_UNROLL for (int i = 0; i < N; i++) o |= v[i];
    return o == 0;
  }
  CPX_HD bool eq(const Fe& b) const {
    uint32_t o = 0;
    CPX_UNROLL for (int i = 0; i < N; i++) o |= v[i] ^ b.v[i];
    return o == 0;
  }
};

// r = a - p if a >= p else a   (a < 2p)
template <class C> CPX_HD void fe_reduce_once(Fe<C>& a) {
  constexpr int N = C::N;
  uint32_t s[N];
  uint32_t borrow = 0;
  CPX_UNROLL for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)a.v[i] - C::P[i] - borrow;
    s[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
  CPX_UNROLL for (int i = 0; i < N; i++) a.v[i] = borrow ? a.v[i] : s[i];
}

template <class C> CPX_HD Fe<C> fe_add(const Fe<C>& a, const Fe<C>& b) {
  constexpr int N = C::N;
  Fe<C> r;
  uint32_t carry = 0;
  CPX_UNROLL for (int i = 0; i < N; i++) {
    uint64_t s = (uint64_t)a.v[i] + b.v[i] + carry;
    r.v[i] = (uint32_t)s;
    carry = (uint32_t)(s >> 32);
  }
  // both moduli leave the top bit of the top limb clear, so a + b < 2^(32N): carry == 0
  fe_reduce_once(r);
  return r;
}
template <class C> CPX_HD Fe<C> fe_sub(const Fe<C>& a, const Fe<C>& b) {
  constexpr int N = C::N;
  Fe<C> r;
  uint32_t borrow = 0;
  CPX_UNROLL for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)a.v[i] - b.v[i] - borrow;
    r.v[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
  uint32_t mask = 0u - borrow, carry = 0;
  CPX_UNROLL for (int i = 0; i < N; i++) {
    uint64_t s = (uint64_t)r.v[i] + (C::P[i] & mask) + carry;
    r.v[i] = (uint32_t)s;
    carry = (uint32_t)(s >> 32);
  }
  return r;
}
template <class C> CPX_HD Fe<C> fe_neg(const Fe<C>& a) {
  constexpr int N = C::N;
  Fe<C> r;
  uint32_t borrow = 0, nz = 0;
  CPX_UNROLL for (int i = 0; i < N; i++) nz |= a.v[i];
  CPX_UNROLL for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)C::P[i] - a.v[i] - borrow;
    r.v[i] = nz ? (uint32_t)d : 0u;
    borrow = (uint32_t)(d >> 63);
  }
  return r;
}
template <class C> CPX_HD Fe<C> fe_dbl(const Fe<C>& a) {
  constexpr int N = C::N;
  Fe<C> r;
  uint32_t carry = 0;
  CPX_UNROLL for (int i = 0; i < N; i++) {
    uint32_t x = a.v[i];
    r.v[i] = (x << 1) | carry;
    carry = x >> 31;
  }
  fe_reduce_once(r);
  return r;
}

// 96-bit column accumulator step: (c2 : acc) += a * b.
// On gfx950: v_mad_u64_u32 (64-bit accumulate, carry-out to VCC) + v_addc_co_u32 (carry into the third
// word) — no register shuffling, which is what the compiler's own lowering of 64-bit carry chains drowns
// in (~1000 v_mov/v_lshl_add_u64 per product).  gfx950 needs 2 wait states between a VALU that writes VCC
// and a VALU that reads it as carry-in; hipcc pads nothing inside an asm string, hence the s_nop 1.
#if defined(__HIP_DEVICE_COMPILE__)
#define CPX_MAC_V(acc, c2, a, b) \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(c2) : "v"(a), "v"(b) : "vcc")
#define CPX_MAC_S(acc, c2, a, b) \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(c2) : "v"(a), "s"(b) : "vcc")
#else
#define CPX_MAC_V(acc, c2, a, b)                     \
  do {                                               \
    uint64_t _p = (uint64_t)(a) * (b);               \
    (acc) += _p;                                     \
    (c2) += ((acc) < _p) ? 1u : 0u;                  \
  } while (0)
#define CPX_MAC_S(acc, c2, a, b) CPX_MAC_V(acc, c2, a, b)
#endif

// Montgomery product a*b*R^-1 mod p — finely integrated product scanning (FIPS): column k gathers
// all a_i*b_(k-i) and m_i*p_(k-i) into a 96-bit accumulator, emits one result limb, shifts by a limb.
// 2N^2 multiply-accumulates, each a (v_mad_u64_u32, v_addc_co_u32) pair; the modulus limbs sit in SGPRs.
#ifndef CPX_FE_MUL_VARIANT
#define CPX_FE_MUL_VARIANT 1
#endif
#if CPX_FE_MUL_VARIANT == 0
// variant 0 (A/B reference): CIOS "no-carry" form, carry chains left to the compiler
template <class C> CPX_HD Fe<C> fe_mul_body(const Fe<C>& a, const Fe<C>& b) {
  constexpr int N = C::N;
  uint32_t t[N];
  CPX_UNROLL for (int i = 0; i < N; i++) t[i] = 0;
  CPX_UNROLL for (int i = 0; i < N; i++) {
    const uint32_t bi = b.v[i];
    uint64_t A = (uint64_t)a.v[0] * bi + t[0];
    const uint32_t t0 = (uint32_t)A;
    A >>= 32;
    const uint32_t m = t0 * C::INV;
    uint64_t Cc = (uint64_t)m * C::P[0] + t0;
    Cc >>= 32;
    CPX_UNROLL for (int j = 1; j < N; j++) {
      A += (uint64_t)a.v[j] * bi + t[j];
      Cc += (uint64_t)m * C::P[j] + (uint32_t)A;
      A >>= 32;
      t[j - 1] = (uint32_t)Cc;
      Cc >>= 32;
    }
    t[N - 1] = (uint32_t)(Cc + A);
  }
  Fe<C> r;
  CPX_UNROLL for (int i = 0; i < N; i++) r.v[i] = t[i];
  fe_reduce_once(r);
  return r;
}
#else
template <class C> CPX_HD Fe<C> fe_mul_body(const Fe<C>& a, const Fe<C>& b) {
  constexpr int N = C::N;
  uint32_t m[N], t[N];
  uint64_t acc = 0;
  uint32_t c2 = 0;
  CPX_UNROLL for (int k = 0; k < N; k++) {
    CPX_UNROLL for (int i = 0; i < k; i++) {
      CPX_MAC_V(acc, c2, a.v[i], b.v[k - i]);
      CPX_MAC_S(acc, c2, m[i], C::P[k - i]);
    }
    CPX_MAC_V(acc, c2, a.v[k], b.v[0]);
    m[k] = (uint32_t)acc * C::INV;
    CPX_MAC_S(acc, c2, m[k], C::P[0]);
    acc = (acc >> 32) | ((uint64_t)c2 << 32);
    c2 = 0;
  }
  CPX_UNROLL for (int k = N; k < 2 * N - 1; k++) {
    CPX_UNROLL for (int i = k - N + 1; i < N; i++) {
      CPX_MAC_V(acc, c2, a.v[i], b.v[k - i]);
      CPX_MAC_S(acc, c2, m[i], C::P[k - i]);
    }
    t[k - N] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)c2 << 32);
    c2 = 0;
  }
  t[N - 1] = (uint32_t)acc;   // the top word is zero: result < 2p < 2^(32N)
  Fe<C> r;
  CPX_UNROLL for (int i = 0; i < N; i++) r.v[i] = t[i];
  fe_reduce_once(r);
  return r;
}
#endif
// Out-of-line entry points.  The AMDGPU calling convention passes at most 16 dwords of aggregate arguments in
// registers; `fe_mul(Fe, Fe)` therefore sent its second operand through scratch memory (store, s_waitcnt
// vmcnt(0), scratch_load) on EVERY multiplication.  Scalar arguments are not subject to that budget, so the
// 2 x N limbs travel as 2N u32 register arguments and the result comes back in N registers.
#define CPX_L12(p) p##0, p##1, p##2, p##3, p##4, p##5, p##6, p##7, p##8, p##9, p##10, p##11
#define CPX_A12(p) uint32_t p##0, uint32_t p##1, uint32_t p##2, uint32_t p##3, uint32_t p##4, uint32_t p##5, uint32_t p##6, uint32_t p##7, uint32_t p##8, uint32_t p##9, uint32_t p##10, uint32_t p##11
#define CPX_L8(p) p##0, p##1, p##2, p##3, p##4, p##5, p##6, p##7
#define CPX_A8(p) uint32_t p##0, uint32_t p##1, uint32_t p##2, uint32_t p##3, uint32_t p##4, uint32_t p##5, uint32_t p##6, uint32_t p##7
CPX_HD_FN Fe<FpCfg> fe_mul_regs12(CPX_A12(a), CPX_A12(b)) {
  const Fe<FpCfg> x{{CPX_L12(a)}}, y{{CPX_L12(b)}};
  return fe_mul_body<FpCfg>(x, y);
}
CPX_HD_FN Fe<FrCfg> fe_mul_regs8(CPX_A8(a), CPX_A8(b)) {
  const Fe<FrCfg> x{{CPX_L8(a)}}, y{{CPX_L8(b)}};
  return fe_mul_body<FrCfg>(x, y);
}
CPX_HD Fe<FpCfg> fe_mul(const Fe<FpCfg>& a, const Fe<FpCfg>& b) {
  return fe_mul_regs12(a.v[0], a.v[1], a.v[2], a.v[3], a.v[4], a.v[5], a.v[6], a.v[7], a.v[8], a.v[9], a.v[10], a.v[11],
                       b.v[0], b.v[1], b.v[2], b.v[3], b.v[4], b.v[5], b.v[6], b.v[7], b.v[8], b.v[9], b.v[10], b.v[11]);
}
CPX_HD Fe<FrCfg> fe_mul(const Fe<FrCfg>& a, const Fe<FrCfg>& b) {
  return fe_mul_regs8(a.v[0], a.v[1], a.v[2], a.v[3], a.v[4], a.v[5], a.v[6], a.v[7],
                      b.v[0], b.v[1], b.v[2], b.v[3], b.v[4], b.v[5], b.v[6], b.v[7]);
}
template <class C> CPX_HD Fe<C> fe_sqr(const Fe<C>& a) { return fe_mul(a, a); }

// Montgomery form -> canonical integer (multiply by 1) and back (multiply by R^2)
template <class C> CPX_HD Fe<C> fe_from_mont(const Fe<C>& a) {
  Fe<C> o = Fe<C>::zero();
  o.v[0] = 1;
  return fe_mul(a, o);
}
template <class C> CPX_HD Fe<C> fe_to_mont(const Fe<C>& a) { return fe_mul(a, Fe<C>::r2()); }

// canonical-integer comparison helper: a > b on raw limbs
template <class C> CPX_HD bool fe_raw_gt(const Fe<C>& a, const Fe<C>& b) {
  constexpr int N = C::N;
  bool gt = false, decided = false;
  CPX_UNROLL for (int i = N - 1; i >= 0; i--) {
    bool g = a.v[i] > b.v[i], l = a.v[i] < b.v[i];
    gt = decided ? gt : g;
    decided = decided || g || l;
  }
  return gt;
}

// a^e for a public exponent given as little-endian 32-bit limbs (nl limbs); plain square-and-multiply.
// Not unrolled: used for inversion / sqrt where the exponent is a fixed constant.
template <class C> CPX_HD Fe<C> fe_pow(const Fe<C>& a, const uint32_t* e, int nl) {
  Fe<C> r = Fe<C>::one();
  bool started = false;
  for (int i = nl * 32 - 1; i >= 0; i--) {
    if (started) r = fe_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1u) {
      r = started ? fe_mul(r, a) : a;
      started = true;
    }
  }
  return r;
}
// Fermat inverse a^(p-2); 0 -> 0.
template <class C> CPX_HD Fe<C> fe_inv(const Fe<C>& a) {
  constexpr int N = C::N;
  uint32_t e[N];
  uint32_t borrow = 2;   // e = p - 2 with borrow propagation (r ends in ...00000001)
  CPX_UNROLL for (int i = 0; i < N; i++) {
    uint64_t d = (uint64_t)C::P[i] - borrow;
    e[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
  return fe_pow(a, e, N);
}

typedef Fe<FpCfg> Fp;
typedef Fe<FrCfg> Fr;

// ---- inversion by the binary extended Euclidean algorithm (one value at a time) ----
// a^-1 mod p for a canonical integer 0 < a < p (plain 12 x u32 words, no Montgomery factor).  About 760 halvings
// and 380 subtractions of 384-bit integers, ~85 k cheap instructions, against the 571 Montgomery products (~290 k
// instructions) of Fermat's a^(p-2): the kernels that invert ONE value per work-group (batch inversion by product
// scans) are bound by the latency of exactly this chain.  Variable time: only public data (group elements of the
// proof) is ever inverted on the device.
CPX_HD void words_inv_mod_p(const uint32_t* a_in, uint32_t* out) {
  constexpr int N = 12;
  uint32_t u[N], v[N], x1[N], x2[N];
  CPX_UNROLL for (int i = 0; i < N; i++) {
    u[i] = a_in[i];
    v[i] = FpCfg::P[i];
    x1[i] = 0;
    x2[i] = 0;
  }
  x1[0] = 1;
  auto is_one = [](const uint32_t* w) {
    uint32_t o = w[0] ^ 1u;
    CPX_UNROLL for (int i = 1; i < N; i++) o |= w[i];
    return o == 0;
  };
  auto geq = [](const uint32_t* a, const uint32_t* b) {
    for (int i = N - 1; i >= 0; i--) {
      if (a[i] != b[i]) return a[i] > b[i];
    }
    return true;
  };
  auto sub = [](uint32_t* a, const uint32_t* b) {   // a -= b, returns the borrow
    uint64_t bw = 0;
    CPX_UNROLL for (int i = 0; i < N; i++) {
      const uint64_t d = (uint64_t)a[i] - b[i] - bw;
      a[i] = (uint32_t)d;
      bw = (d >> 32) & 1u;
    }
    return (uint32_t)bw;
  };
  auto sub_mod = [&](uint32_t* a, const uint32_t* b) {   // a = a - b mod p, both in [0, p)
    if (sub(a, b)) {
      uint64_t c = 0;
      CPX_UNROLL for (int i = 0; i < N; i++) {
        c += (uint64_t)a[i] + FpCfg::P[i];
        a[i] = (uint32_t)c;
        c >>= 32;
      }
    }
  };
  // k trailing zero bits at once (1 <= k <= 31): w >>= k;  x = x / 2^k mod p by adding the multiple m p that clears the
  // low k bits (m = x * (-p^-1) mod 2^k), then shifting — one multiply-accumulate pass instead of k add-and-halve passes
  auto shift_k = [](uint32_t* w, uint32_t top, int k) {   // (top:w) >> k
    CPX_UNROLL for (int i = 0; i < N - 1; i++) w[i] = (w[i] >> k) | (w[i + 1] << (32 - k));
    w[N - 1] = (w[N - 1] >> k) | (top << (32 - k));
  };
  auto strip = [&](uint32_t* w, uint32_t* x) {
    const int k = __builtin_ctz(w[0] | 0x80000000u);      // low word 0: take 31 bits now, the loop comes back for the rest
    shift_k(w, 0, k);
    const uint32_t m = (x[0] * FpCfg::INV) & ((1u << k) - 1u);
    uint64_t c = 0;
    CPX_UNROLL for (int i = 0; i < N; i++) {
      c += (uint64_t)m * FpCfg::P[i] + x[i];
      x[i] = (uint32_t)c;
      c >>= 32;
    }
    shift_k(x, (uint32_t)c, k);
    return k;
  };
  // u + v loses at least one bit per halving: <= 2 * 384 halved bits in total.  The budget only matters for the
  // invalid input a = 0 (never produced by the callers), which must not spin forever on a GPU.
  int budget = 4 * 384;
  while (!is_one(u) && !is_one(v) && budget > 0) {
    while (!(u[0] & 1u) && budget > 0) budget -= strip(u, x1);
    while (!(v[0] & 1u) && budget > 0) budget -= strip(v, x2);
    budget--;
    if (geq(u, v)) {
      sub(u, v);
      sub_mod(x1, x2);
    } else {
      sub(v, u);
      sub_mod(x2, x1);
    }
  }
  const bool first = is_one(u);
  CPX_UNROLL for (int i = 0; i < N; i++) out[i] = first ? x1[i] : x2[i];
}

// Montgomery-form inverse through it: X = a R  ->  X^-1 = a^-1 R^-1  ->  times R^3 (one Montgomery product) = a^-1 R
CPX_HD Fp fe_inv_euclid(const Fp& a) {
  static constexpr uint32_t R3[12] = {0xd94ca1e0u, 0xed48ac6bu, 0x03a7adf8u, 0x315f831eu, 0x615e29ddu, 0x9a53352au,
                                      0x921e1761u, 0x34c04e5eu, 0x65724728u, 0x2512d435u, 0x91755d4du, 0x0aa63460u};
  Fp t, r3;
  words_inv_mod_p(a.v, t.v);
  CPX_UNROLL for (int i = 0; i < 12; i++) r3.v[i] = R3[i];
  return fe_mul(t, r3);
}

}  // namespace cpx
