// BLS12-381 G1 points in the standard form (32-bit-limb Montgomery Fp) — product code (host + gfx950 device).
// The group law itself runs on the table form (g1_28.hpp); this header holds the wire types, the affine
// normalisation step and the compressed encoding.
//
// Points cross the C-ABI in arkworks' in-memory limb layout (include/cpx.h):
//   Aff : x || y                (96 B; the identity is encoded x = y = 0 — (0,0) is not on y^2 = x^3 + 4)
//   Jac : X || Y || Z Jacobian  (144 B; Z = 0 is the identity)
#pragma once
#include "mont32.hpp"

// Point formulas (g1_28.hpp) are inlined into the kernels so that a kernel's __launch_bounds__ governs their
// register budget (a non-kernel function is compiled for the maximum budget, which pins every caller at 2 waves
// per SIMD).
#define CPX_PT_FN CPX_HD

namespace cpx {

struct Aff {
  Fp x, y;
  CPX_HD bool is_identity() const { return x.is_zero() && y.is_zero(); }
  static CPX_HD Aff identity() { return Aff{Fp::zero(), Fp::zero()}; }
};
struct Jac {
  Fp x, y, z;
  CPX_HD bool is_identity() const { return z.is_zero(); }
  static CPX_HD Jac identity() { return Jac{Fp::one(), Fp::one(), Fp::zero()}; }
  static CPX_HD Jac from_affine(const Aff& a) {
    Jac r;
    const bool inf = a.is_identity();
    r.x = a.x;
    r.y = a.y;
    r.z = inf ? Fp::zero() : Fp::one();
    if (inf) {
      r.x = Fp::one();
      r.y = Fp::one();
    }
    return r;
  }
};

CPX_HD Aff aff_neg(const Aff& p) { return Aff{p.x, fe_neg(p.y)}; }
CPX_HD Aff aff_cneg(const Aff& p, bool neg) {
  Aff r = p;
  Fp ny = fe_neg(p.y);
  CPX_UNROLL for (int i = 0; i < Fp::N; i++) r.y.v[i] = neg ? ny.v[i] : p.y.v[i];
  return r;
}
CPX_HD Jac jac_neg(const Jac& p) { return Jac{p.x, fe_neg(p.y), p.z}; }

// Jacobian -> affine given zinv = Z^-1 (caller handles identity)
CPX_HD Aff jac_to_affine_with_zinv(const Jac& p, const Fp& zinv) {
  Fp zi2 = fe_sqr(zinv);
  Aff r;
  r.x = fe_mul(p.x, zi2);
  r.y = fe_mul(fe_mul(p.y, zi2), zinv);
  return r;
}

// zcash / ark-bls12-381 compressed encoding of an affine point (48 bytes, big-endian x,
// bit7 = compressed, bit6 = infinity, bit5 = y lexicographically largest i.e. y > -y), as twelve 32-bit words whose
// little-endian memory image is the encoding: a kernel stores them with three 16-byte stores.
CPX_HD uint32_t bswap32(uint32_t w) { return (w >> 24) | ((w >> 8) & 0xff00u) | ((w << 8) & 0xff0000u) | (w << 24); }
CPX_HD void aff_compress_words(const Aff& p, uint32_t w[12]) {
  if (p.is_identity()) {
    w[0] = 0xc0u;   // byte 0
    CPX_UNROLL for (int i = 1; i < 12; i++) w[i] = 0;
    return;
  }
  const Fp xc = fe_from_mont(p.x), yc = fe_from_mont(p.y);
  // y > -y  <=>  y > p - y (canonical integers; y != 0 on this curve: there is no point of order two)
  Fp pm, nyc;
  CPX_UNROLL for (int i = 0; i < 12; i++) pm.v[i] = FpCfg::P[i];
  {
    uint32_t borrow = 0;
    CPX_UNROLL for (int i = 0; i < 12; i++) {
      const uint64_t d = (uint64_t)pm.v[i] - yc.v[i] - borrow;
      nyc.v[i] = (uint32_t)d;
      borrow = (uint32_t)(d >> 63);
    }
  }
  uint32_t top = xc.v[11] | 0x80000000u;
  if (fe_raw_gt(yc, nyc)) top |= 0x20000000u;
  w[0] = bswap32(top);
  CPX_UNROLL for (int l = 1; l < 12; l++) w[l] = bswap32(xc.v[11 - l]);
}
CPX_HD void aff_compress(const Aff& p, uint8_t* out) {
  uint32_t w[12];
  aff_compress_words(p, w);
  CPX_UNROLL for (int l = 0; l < 12; l++) {
    out[4 * l + 0] = (uint8_t)w[l];
    out[4 * l + 1] = (uint8_t)(w[l] >> 8);
    out[4 * l + 2] = (uint8_t)(w[l] >> 16);
    out[4 * l + 3] = (uint8_t)(w[l] >> 24);
  }
}

// Signed-digit (NAF-like, width w) recoding helpers operate on canonical 256-bit scalars held as
// 8 x u32 little-endian limbs.
CPX_HD uint32_t scalar_bits(const uint32_t* k, int pos, int width) {   // bits [pos, pos+width), pos+width <= 256+
  int limb = pos >> 5, off = pos & 31;
  uint64_t w = k[limb];
  if (limb + 1 < 8) w |= (uint64_t)k[limb + 1] << 32;
  return (uint32_t)(w >> off) & ((1u << width) - 1u);
}

}  // namespace cpx
