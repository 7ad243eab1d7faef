// Modular inversion mod the BLS12-381 base field prime by batched division steps — product code, host + device.
//
// The algorithm of D. J. Bernstein and B.-Y. Yang ("Fast constant-time gcd computation and modular inversion", 2019)
// in the variable-time 30-bit form P. Wuille derived for libsecp256k1 (modinv32): 30 division steps are run on the
// low 30 bits of (f, g) only — plain 32-bit integer work — while their effect is accumulated in a 2 x 2 matrix with
// entries <= 2^30; the matrix is then applied once to the full-length (f, g) and to the Bezout pair (d, e), the
// latter modulo p.  About 25 batches of (~10 cheap iterations + 2 x 13 limb passes) instead of the ~500 full-length
// subtract / shift iterations of the binary Euclidean algorithm: ~5x fewer instructions on one GPU lane, which is
// what the latency of the batch-inversion kernels (k_finalize_ranges, k_table_build, ...) consists of.
// Variable time: only public values (group elements of proofs) are inverted with it.
//
// The division-step / matrix-update structure follows libsecp256k1's src/modinv32_impl.h, which is distributed under
// the MIT license:
//   Copyright (c) 2020 Peter Dettman, Pieter Wuille
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without limitation the
//   rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to permit
//   persons to whom the Software is furnished to do so, subject to the following conditions:
//   The above copyright notice and this permission notice shall be included in all copies or substantial portions of the
//   Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE
//   WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT.  IN NO EVENT SHALL THE AUTHORS OR
//   COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR
//   OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
// The code below is a re-derivation for a 381-bit modulus on 13 limbs (different limb count, modulus constants and
// final normalisation), written for this repository.
#pragma once
#include "mont32.hpp"

namespace cpx {

template <int NL> struct Signed30T {   // NL signed limbs of 30 bits, value = sum v[i] 2^(30 i); limbs 0..NL-2 in [0, 2^30), the last one carries the sign
  int32_t v[NL];
};
typedef Signed30T<13> Signed30;
struct Trans2x2 {
  int32_t u, v, q, r;
};

struct ModInv30Cfg {   // BLS12-381 base field: 381 bits = 12 x u32 words = 13 limbs
  static constexpr int NL = 13, NW = 12, MAX_BATCHES = 40;
  static constexpr int32_t M30 = (1 << 30) - 1;
  static constexpr int32_t P30[13] = {0x3fffaaab, 0x27fbffff, 0x153ffffb, 0x2affffac, 0x30f6241e, 0x034a83da, 0x112bf673,
                                      0x12e13ce1, 0x2cd76477, 0x1ed90d2e, 0x29a4b1ba, 0x3a8e5ff9, 0x001a0111};
  static constexpr uint32_t PINV30 = 0x00030003;   // p^-1 mod 2^30
};

struct ModInv30FrCfg {   // BLS12-381 scalar field: 255 bits = 8 x u32 words = 9 limbs (the Fr inversions of the device-side protocol steps)
  static constexpr int NL = 9, NW = 8, MAX_BATCHES = 28;   // <= 735 division steps for 255-bit inputs (the paper's bound), 30 per batch
  static constexpr int32_t M30 = (1 << 30) - 1;
  static constexpr int32_t P30[9] = {0x00000001, 0x3ffffffc, 0x3fe5bfef, 0x2f6900bf, 0x21d80553, 0x27602026, 0x17d48333, 0x29d4ca67, 0x000073ed};
  static constexpr uint32_t PINV30 = 0x00000001;   // r^-1 mod 2^30
};

// 30 division steps on the low bits; returns the new eta (= -delta) and the transition matrix t with
// t * [f, g] = 2^30 * [f', g'].
CPX_HD int32_t modinv30_divsteps(int32_t eta, uint32_t f0, uint32_t g0, Trans2x2& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
  uint32_t f = f0, g = g0;
  int i = 30;
  for (;;) {
    const int zeros = __builtin_ctz(g | (0xffffffffu << i));   // trailing zeros of g, at most i
    g >>= zeros;
    u <<= zeros;
    v <<= zeros;
    eta -= zeros;
    i -= zeros;
    if (i == 0) break;
    if (eta < 0) {   // swap: (f, g) <- (g, -f)
      eta = -eta;
      uint32_t tmp = f;
      f = g;
      g = 0u - tmp;
      tmp = u;
      u = q;
      q = 0u - tmp;
      tmp = v;
      v = r;
      r = 0u - tmp;
    }
    // cancel up to min(eta + 1, i, 8) low bits of g with a multiple w of f:  w = -g / f mod 2^limit
    const int limit = (eta + 1) > i ? i : (eta + 1);
    const uint32_t m = (0xffffffffu >> (32 - limit)) & 255u;
    uint32_t fi = f;                 // f^-1 mod 2^3 (f odd) -> 2^6 -> 2^12 by Newton steps
    fi *= 2u - f * fi;
    fi *= 2u - f * fi;
    const uint32_t w = ((0u - fi) * g) & m;
    g += f * w;
    q += u * w;
    r += v * w;
  }
  t.u = (int32_t)u;
  t.v = (int32_t)v;
  t.q = (int32_t)q;
  t.r = (int32_t)r;
  return eta;
}

// [d, e] <- t * [d, e] / 2^30 mod p, both kept in (-2p, p)
template <class C> CPX_HD void modinv30_update_de(Signed30T<C::NL>& d, Signed30T<C::NL>& e, const Trans2x2& t) {
  constexpr int32_t M30 = C::M30;
  constexpr int NL = C::NL;
  const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int32_t sd = d.v[NL - 1] >> 31, se = e.v[NL - 1] >> 31;   // all-ones if negative
  int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
  int32_t di = d.v[0], ei = e.v[0];
  int64_t cd = (int64_t)u * di + (int64_t)v * ei, ce = (int64_t)q * di + (int64_t)r * ei;
  // choose the multiples of p that clear the low 30 bits of the two sums
  md -= (int32_t)((C::PINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
  me -= (int32_t)((C::PINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd += (int64_t)C::P30[0] * md;
  ce += (int64_t)C::P30[0] * me;
  cd >>= 30;
  ce >>= 30;
  CPX_UNROLL for (int i = 1; i < NL; i++) {
    di = d.v[i];
    ei = e.v[i];
    cd += (int64_t)u * di + (int64_t)v * ei;
    ce += (int64_t)q * di + (int64_t)r * ei;
    cd += (int64_t)C::P30[i] * md;
    ce += (int64_t)C::P30[i] * me;
    d.v[i - 1] = (int32_t)cd & M30;
    cd >>= 30;
    e.v[i - 1] = (int32_t)ce & M30;
    ce >>= 30;
  }
  d.v[NL - 1] = (int32_t)cd;
  e.v[NL - 1] = (int32_t)ce;
}

// [f, g] <- t * [f, g] / 2^30 (exact)
template <class C> CPX_HD void modinv30_update_fg(Signed30T<C::NL>& f, Signed30T<C::NL>& g, const Trans2x2& t) {
  constexpr int32_t M30 = C::M30;
  constexpr int NL = C::NL;
  const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
  int32_t fi = f.v[0], gi = g.v[0];
  int64_t cf = (int64_t)u * fi + (int64_t)v * gi, cg = (int64_t)q * fi + (int64_t)r * gi;
  cf >>= 30;
  cg >>= 30;
  CPX_UNROLL for (int i = 1; i < NL; i++) {
    fi = f.v[i];
    gi = g.v[i];
    cf += (int64_t)u * fi + (int64_t)v * gi;
    cg += (int64_t)q * fi + (int64_t)r * gi;
    f.v[i - 1] = (int32_t)cf & M30;
    cf >>= 30;
    g.v[i - 1] = (int32_t)cg & M30;
    cg >>= 30;
  }
  f.v[NL - 1] = (int32_t)cf;
  g.v[NL - 1] = (int32_t)cg;
}

// a^-1 mod p for a canonical integer 0 < a < p given as C::NW x u32 words; out likewise.  a = 0 gives 0.
template <class C> CPX_HD void words_inv_divsteps(const uint32_t* a, uint32_t* out) {
  constexpr int32_t M30 = C::M30;
  constexpr int NL = C::NL, NW = C::NW;
  Signed30T<NL> d, e, f, g;
  CPX_UNROLL for (int i = 0; i < NL; i++) {
    d.v[i] = 0;
    e.v[i] = 0;
    f.v[i] = C::P30[i];
    const int bit = 30 * i, w = bit >> 5, o = bit & 31;
    uint64_t x = w < NW ? a[w] : 0;
    if (w + 1 < NW) x |= (uint64_t)a[w + 1] << 32;
    g.v[i] = (int32_t)((x >> o) & (uint32_t)M30);
  }
  e.v[0] = 1;
  int32_t eta = -1;
  // every batch performs 30 division steps; at most ~1100 are ever needed for 381-bit inputs (the bound of the paper),
  // the loop stops as soon as g = 0
  for (int batch = 0; batch < C::MAX_BATCHES; batch++) {
    Trans2x2 t;
    eta = modinv30_divsteps(eta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    modinv30_update_de<C>(d, e, t);
    modinv30_update_fg<C>(f, g, t);
    int32_t nz = 0;
    CPX_UNROLL for (int i = 0; i < NL; i++) nz |= g.v[i];
    if (nz == 0) break;
  }
  // f = +-1 now (gcd); d = +-a^-1 in (-2p, p): fix the sign, then bring into [0, p)
  const int32_t fneg = f.v[NL - 1] >> 31;   // all-ones if f = -1
  int32_t carry = 0;
  CPX_UNROLL for (int i = 0; i < NL; i++) {   // d <- (f < 0 ? -d : d), limbs renormalised
    int32_t x = (d.v[i] ^ fneg) - fneg + carry;
    carry = x >> 30;
    d.v[i] = i < NL - 1 ? (x & M30) : x;
  }
  for (int round = 0; round < 2; round++) {   // d in (-2p, 2p): add p while negative
    const int32_t neg = d.v[NL - 1] >> 31;
    carry = 0;
    CPX_UNROLL for (int i = 0; i < NL; i++) {
      int32_t x = d.v[i] + (C::P30[i] & neg) + carry;
      carry = x >> 30;
      d.v[i] = i < NL - 1 ? (x & M30) : x;
    }
  }
  {   // subtract p if d >= p
    Signed30T<NL> s;
    carry = 0;
    CPX_UNROLL for (int i = 0; i < NL; i++) {
      int32_t x = d.v[i] - C::P30[i] + carry;
      carry = x >> 30;
      s.v[i] = i < NL - 1 ? (x & M30) : x;
    }
    const int32_t keep = s.v[NL - 1] >> 31;   // negative: d < p, keep d
    CPX_UNROLL for (int i = 0; i < NL; i++) d.v[i] = (d.v[i] & keep) | (s.v[i] & ~keep);
  }
  CPX_UNROLL for (int k = 0; k < NW; k++) {   // 30-bit limbs -> 32-bit words
    const int bit = 32 * k, i = bit / 30, o = bit % 30;
    uint64_t x = (uint64_t)(uint32_t)d.v[i] >> o;
    if (i + 1 < NL) x |= (uint64_t)(uint32_t)d.v[i + 1] << (30 - o);
    if (i + 2 < NL && 60 - o < 32) x |= (uint64_t)(uint32_t)d.v[i + 2] << (60 - o);
    out[k] = (uint32_t)x;
  }
}

CPX_HD void words_inv_mod_p_divsteps(const uint32_t* a, uint32_t* out) { words_inv_divsteps<ModInv30Cfg>(a, out); }

// Montgomery-form inverse through it: X = a R  ->  X^-1 = a^-1 R^-1  ->  times R^3 (one Montgomery product) = a^-1 R
CPX_HD Fp fe_inv_divsteps(const Fp& a) {
  const uint32_t R3[12] = {0xd94ca1e0u, 0xed48ac6bu, 0x03a7adf8u, 0x315f831eu, 0x615e29ddu, 0x9a53352au,
                           0x921e1761u, 0x34c04e5eu, 0x65724728u, 0x2512d435u, 0x91755d4du, 0x0aa63460u};
  Fp t, r3;
  words_inv_mod_p_divsteps(a.v, t.v);
  CPX_UNROLL for (int i = 0; i < 12; i++) r3.v[i] = R3[i];
  return fe_mul(t, r3);
}

// The same for the scalar field (Montgomery radix 2^256): the device-side protocol steps invert challenges with it.  0 -> 0.
CPX_HD Fr fr_inv_divsteps(const Fr& a) {
  const uint32_t R3[8] = {0x439b73afu, 0xc62c1807u, 0x8cf06990u, 0x1b3e0d18u, 0xc7b5f418u, 0x73d13c71u, 0xc8db33e9u, 0x6e2a5bb9u};   // R^3 mod r
  Fr t, r3;
  words_inv_divsteps<ModInv30FrCfg>(a.v, t.v);
  CPX_UNROLL for (int i = 0; i < 8; i++) r3.v[i] = R3[i];
  return fe_mul(t, r3);
}

}  // namespace cpx
