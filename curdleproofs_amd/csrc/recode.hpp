// Scalar recodings shared by the gfx950 kernels and the host-side emulation tests.
#pragma once
#include "mont32.hpp"
#include "glv.hpp"

namespace cpx {

// Signed radix-16 digits of a canonical scalar k < 2^255 (8 x u32 LE): k = sum d_w 16^w,
// d_w in [-7, 8], 64 digits.  Digit magnitudes 1..8 index the 8 Pippenger buckets of a window.
CPX_HD void recode_signed16(const uint32_t* k, int8_t* d /*64*/, int stride = 1) {
  uint32_t carry = 0;
  for (int w = 0; w < 64; w++) {
    uint32_t nib = (k[w >> 3] >> ((w & 7) * 4)) & 0xfu;
    uint32_t v = nib + carry;                 // 0..16
    int dv = (v > 8) ? (int)v - 16 : (int)v;  // -7..8
    carry = (v > 8) ? 1u : 0u;
    d[w * stride] = (int8_t)dv;
  }
}

// Non-adjacent form of a canonical scalar k < 2^256: k = sum_{i=0}^{256} e_i 2^i, e_i in {-1,0,1},
// as two 288-bit masks (9 x u32 each): nz bit i = (e_i != 0), ng bit i = (e_i < 0).
CPX_HD void recode_naf(const uint32_t* k, uint32_t* nz /*9*/, uint32_t* ng /*9*/) {
  uint32_t x[9];
  for (int i = 0; i < 8; i++) x[i] = k[i];
  x[8] = 0;
  for (int i = 0; i < 9; i++) nz[i] = ng[i] = 0;
  // classic: while x: if x odd: e = 2 - (x mod 4); x -= e;  x >>= 1
  // implemented bitwise: 3x = x + 2x; naf nonzero = (3x ^ x) >> 1 ; sign from 3x bits
  // np = 3x, nm = x :  e_i = bit_{i+1}(3x) - bit_{i+1}(x)
  uint32_t x3[10];
  uint64_t c = 0;
  for (int i = 0; i < 9; i++) {
    uint64_t twice = ((uint64_t)x[i] << 1) | (i ? (x[i - 1] >> 31) : 0);
    uint64_t s = (uint64_t)x[i] + (uint32_t)twice + c;
    x3[i] = (uint32_t)s;
    c = s >> 32;
  }
  x3[9] = (uint32_t)c;
  for (int i = 0; i < 9; i++) {
    // shift right by one: bit i of result = bit i+1 of source
    uint32_t p = (x3[i] >> 1) | (x3[i + 1] << 31);
    uint32_t m = (x[i] >> 1) | ((i < 8 ? x[i + 1] : 0u) << 31);
    nz[i] = p ^ m;
    ng[i] = m & ~p;
  }
}

// ---- per-element scalar multiplication (k_smul): the endomorphism split of glv.hpp, k = +-(+-|t| + q N), N P = (beta x, -y), and a
// non-adjacent form of each half: k P = sum_{i=0}^{128} (e_i P + f_i N P) 2^i, e_i, f_i in {-1, 0, 1} — 129 doublings and ~86
// additions instead of 257 and ~86.  nz / ng: digit non-zero / negative (the signs of k and t folded in), bit i of word i / 32;
// half 0 = the digits of P, half 1 = the digits of N P.
struct SmulNaf {
  uint32_t nz[2][5], ng[2][5];
};
CPX_HD void recode_smul_glv(const uint32_t* k /*canonical, 8 words*/, SmulNaf& o) {
  uint32_t t[8], q[8], nk, nt, nz[9], ng[9];
  for (int i = 4; i < 8; i++) t[i] = q[i] = 0;
  glv_split(k, t, q, nk, nt);
  recode_naf(t, nz, ng);
  for (int i = 0; i < 5; i++) {
    o.nz[0][i] = nz[i];
    o.ng[0][i] = (nk ^ nt) ? (ng[i] ^ nz[i]) : ng[i];
  }
  recode_naf(q, nz, ng);
  for (int i = 0; i < 5; i++) {
    o.nz[1][i] = nz[i];
    o.ng[1][i] = nk ? (ng[i] ^ nz[i]) : ng[i];
  }
}

// ---- fixed-base windows (k_msm_fix, k_late_fix): signed radix-2^CB digits of a canonical scalar k < 2^255 ----
// k = sum_w d_w 2^(CB w),  d_w in [-2^(CB-1), 2^(CB-1) - 1],  W = ceil(256 / CB) windows (CB = 8: 32, 16: 16, 19: 14 covering 266 bits);
// |d| indexes the table of multiples (entry |d| - 1).  No carry leaves the top window: k < r < 2^255.
template <int CB> struct FixWin {
  static constexpr int W = (256 + CB - 1) / CB;
  static constexpr uint32_t HALF = 1u << (CB - 1), FULL = 1u << CB;
};
// digit stream from a shift register: next() returns d_0, d_1, ... (rolled loops with static register indices on the device)
template <int CB> struct FixDigitStream {
  uint32_t v[8];
  uint32_t carry;
  CPX_HD explicit FixDigitStream(const uint32_t* k) : carry(0) {
    CPX_UNROLL for (int j = 0; j < 8; j++) v[j] = k[j];
  }
  CPX_HD int next() {
    const uint32_t x = (v[0] & (FixWin<CB>::FULL - 1)) + carry;
    CPX_UNROLL for (int j = 0; j < 7; j++) v[j] = (v[j] >> CB) | (v[j + 1] << (32 - CB));
    v[7] >>= CB;
    carry = (x >= FixWin<CB>::HALF) ? 1u : 0u;
    return (x >= FixWin<CB>::HALF) ? (int)x - (int)FixWin<CB>::FULL : (int)x;
  }
};
// the NW digits of windows [w0, w0 + NW) written to out[(w - w0) * stride] (DT: int16_t up to CB = 16, int32_t above)
template <int CB, int NW, class DT> CPX_HD void fix_window_digits(const uint32_t* k, int w0, DT* out, int stride) {
  uint32_t carry = 0;
  for (int w = 0; w < FixWin<CB>::W; w++) {
    const int bit = w * CB, wd = bit >> 5;
    const uint64_t two = (uint64_t)k[wd] | ((uint64_t)(wd + 1 < 8 ? k[wd + 1] : 0u) << 32);   // a chunk may straddle two words
    const uint32_t x = ((uint32_t)(two >> (bit & 31)) & (FixWin<CB>::FULL - 1)) + carry;
    const int d = (x >= FixWin<CB>::HALF) ? (int)x - (int)FixWin<CB>::FULL : (int)x;
    carry = (x >= FixWin<CB>::HALF) ? 1u : 0u;
    if (w >= w0 && w < w0 + NW) out[(w - w0) * stride] = (DT)d;
  }
}

// ---- late rounds (k_late_msm): signed radix-16 digits in [-7, 8] of a value v < 2^127 (an endomorphism half), biased by 7 and
// packed eight per word: v = sum_j (nibble_j - 7) 16^j, j < 32; no carry leaves nibble 31 ----
CPX_HD void recode_signed_nibbles_biased(const uint32_t* v /*4*/, uint32_t* packed /*4*/) {
  uint32_t carry = 0;
  CPX_UNROLL for (int wd = 0; wd < 4; wd++) {
    packed[wd] = 0;
    CPX_UNROLL for (int q = 0; q < 8; q++) {
      const uint32_t x = ((v[wd] >> (4 * q)) & 15u) + carry;   // 0 .. 16
      const int d = x > 8 ? (int)x - 16 : (int)x;
      carry = x > 8 ? 1u : 0u;
      packed[wd] |= (uint32_t)(d + 7) << (4 * q);
    }
  }
}

}  // namespace cpx
