// BLS12-381 G1 group law on 32-bit-limb Montgomery Fp — product code (host + gfx950 device).
//
// Points cross the C-ABI in arkworks' in-memory limb layout (include/cpx.h):
//   Aff : x || y                (96 B; the identity is encoded x = y = 0 — (0,0) is not on y^2 = x^3 + 4)
//   Jac : X || Y || Z Jacobian  (144 B; Z = 0 is the identity)
// The formulas are complete w.r.t. the corner cases the reference's inputs really contain:
// identity operands (the G1Affine::zero() padding of curdleproofs.rs:141-155 in /root/reference),
// P + P and P + (-P) (ark-ec's add falls back to doubling / identity the same way).
#pragma once
#include "mont32.cuh"

// Point formulas: inlined into the kernels by default so that a kernel's __launch_bounds__ governs their
// register budget (a non-kernel function is compiled for the maximum budget: 248 VGPRs here, which pins
// every caller at 2 waves per SIMD).  -DCPX_PT_NOINLINE restores real calls for A/B runs.
#if defined(CPX_PT_NOINLINE)
#define CPX_PT_FN CPX_HD_FN
#else
#define CPX_PT_FN CPX_HD
#endif

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

// dbl-2009-l (a = 0): 2M + 5S
CPX_PT_FN Jac jac_dbl(const Jac& p) {
  // Z = 0 stays 0 through the formula (Z3 = 2*Y*Z), so no identity branch is needed.
  Fp A = fe_sqr(p.x), B = fe_sqr(p.y), C = fe_sqr(B);
  Fp D = fe_sub(fe_sub(fe_sqr(fe_add(p.x, B)), A), C);
  D = fe_dbl(D);
  Fp E = fe_add(fe_dbl(A), A), F = fe_sqr(E);
  Jac r;
  r.x = fe_sub(F, fe_dbl(D));
  Fp C8 = fe_dbl(fe_dbl(fe_dbl(C)));
  r.y = fe_sub(fe_mul(E, fe_sub(D, r.x)), C8);
  r.z = fe_dbl(fe_mul(p.y, p.z));
  return r;
}

// madd-2007-bl: 7M + 4S; complete
CPX_PT_FN Jac jac_add_mixed(const Jac& p, const Aff& q) {
  if (q.is_identity()) return p;
  if (p.is_identity()) return Jac::from_affine(q);
  Fp z1z1 = fe_sqr(p.z);
  Fp u2 = fe_mul(q.x, z1z1);
  Fp s2 = fe_mul(fe_mul(q.y, p.z), z1z1);
  Fp h = fe_sub(u2, p.x);
  Fp rr = fe_sub(s2, p.y);
  if (h.is_zero()) {
    if (rr.is_zero()) return jac_dbl(p);
    return Jac::identity();
  }
  rr = fe_dbl(rr);
  Fp hh = fe_sqr(h);
  Fp i = fe_dbl(fe_dbl(hh));
  Fp j = fe_mul(h, i);
  Fp v = fe_mul(p.x, i);
  Jac r;
  r.x = fe_sub(fe_sub(fe_sqr(rr), j), fe_dbl(v));
  r.y = fe_sub(fe_mul(rr, fe_sub(v, r.x)), fe_dbl(fe_mul(p.y, j)));
  r.z = fe_sub(fe_sub(fe_sqr(fe_add(p.z, h)), z1z1), hh);
  return r;
}

// add-2007-bl: 11M + 5S; complete
CPX_PT_FN Jac jac_add(const Jac& p, const Jac& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  Fp z1z1 = fe_sqr(p.z), z2z2 = fe_sqr(q.z);
  Fp u1 = fe_mul(p.x, z2z2), u2 = fe_mul(q.x, z1z1);
  Fp s1 = fe_mul(fe_mul(p.y, q.z), z2z2), s2 = fe_mul(fe_mul(q.y, p.z), z1z1);
  Fp h = fe_sub(u2, u1);
  Fp rr = fe_sub(s2, s1);
  if (h.is_zero()) {
    if (rr.is_zero()) return jac_dbl(p);
    return Jac::identity();
  }
  rr = fe_dbl(rr);
  Fp i = fe_sqr(fe_dbl(h));
  Fp j = fe_mul(h, i);
  Fp v = fe_mul(u1, i);
  Jac r;
  r.x = fe_sub(fe_sub(fe_sqr(rr), j), fe_dbl(v));
  r.y = fe_sub(fe_mul(rr, fe_sub(v, r.x)), fe_dbl(fe_mul(s1, j)));
  r.z = fe_mul(fe_sub(fe_sub(fe_sqr(fe_add(p.z, q.z)), z1z1), z2z2), h);
  return r;
}

// Real-call versions of the three formulas.  k_smul uses these: with the formulas inlined, hipcc (ROCm 7.2)
// miscompiles that kernel at register budgets above 128 VGPRs (every fold/scale parity test fails, with
// or without the inline-asm multiply) while the called form is bit-exact; see DESIGN.md "toolchain notes".
CPX_HD_FN Jac jac_dbl_fn(const Jac& p) { return jac_dbl(p); }
CPX_HD_FN Jac jac_add_mixed_fn(const Jac& p, const Aff& q) { return jac_add_mixed(p, q); }
CPX_HD_FN Jac jac_add_fn(const Jac& p, const Jac& q) { return jac_add(p, q); }

// Jacobian -> affine given zinv = Z^-1 (caller handles identity)
CPX_HD Aff jac_to_affine_with_zinv(const Jac& p, const Fp& zinv) {
  Fp zi2 = fe_sqr(zinv);
  Aff r;
  r.x = fe_mul(p.x, zi2);
  r.y = fe_mul(fe_mul(p.y, zi2), zinv);
  return r;
}

// zcash / ark-bls12-381 compressed encoding of an affine point (48 bytes, big-endian x,
// bit7 = compressed, bit6 = infinity, bit5 = y lexicographically largest i.e. y > -y).
CPX_HD void aff_compress(const Aff& p, uint8_t* out) {
  if (p.is_identity()) {
    out[0] = 0xc0;
    for (int i = 1; i < 48; i++) out[i] = 0;
    return;
  }
  Fp xc = fe_from_mont(p.x), yc = fe_from_mont(p.y), nyc = fe_from_mont(fe_neg(p.y));
  CPX_UNROLL for (int l = 0; l < 12; l++) {
    uint32_t w = xc.v[11 - l];
    out[4 * l + 0] = (uint8_t)(w >> 24);
    out[4 * l + 1] = (uint8_t)(w >> 16);
    out[4 * l + 2] = (uint8_t)(w >> 8);
    out[4 * l + 3] = (uint8_t)w;
  }
  out[0] |= 0x80;
  if (fe_raw_gt(yc, nyc)) out[0] |= 0x20;
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
