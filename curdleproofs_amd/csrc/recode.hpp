// Scalar recodings shared by the gfx950 kernels and the host-side emulation tests.
#pragma once
#include "mont32.hpp"

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

}  // namespace cpx
