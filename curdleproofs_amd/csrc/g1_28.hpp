// BLS12-381 G1 on the carry-free 28-bit-limb field (fp28.hpp) — product code, host + device.
// Same formulas as g1.hpp, arranged so that every quantity that must be tested for zero is a PRODUCT
// (Z3 = 2 * (Z1 * H) instead of (Z1 + H)^2 - Z1Z1 - HH), because a lazy value can only be zero-tested cheaply
// right after a Montgomery multiplication.
// Magnitude bookkeeping (units of p, |.|): stored coordinates stay within |X|,|Y| <= 15.4, |Z| <= 3.6; the
// largest multiplication operands are rr (<= 34.2) squared = 1170 < 2^11.3 = 2521, so every product is in range.
#pragma once
#include "fp28.hpp"
#include "g1.hpp"

namespace cpx {

struct Aff28 {
  F28 x, y;
  CPX_HD bool is_identity() const { return x.is_zero_exact() && y.is_zero_exact(); }
  static CPX_HD Aff28 identity() { return Aff28{F28::zero(), F28::zero()}; }
};
struct Jac28 {
  F28 x, y, z;
  CPX_HD bool is_identity() const { return z.is_zero_exact(); }
  static CPX_HD Jac28 identity() { return Jac28{f28_one(), f28_one(), F28::zero()}; }
  static CPX_HD Jac28 from_affine(const Aff28& a) {
    if (a.is_identity()) return identity();
    return Jac28{a.x, a.y, f28_one()};
  }
};

CPX_HD Aff28 aff28_cneg(const Aff28& p, bool neg) { return Aff28{p.x, f28_cneg(p.y, neg)}; }
// for a point that goes straight into a mixed addition (y only feeds products there)
CPX_HD Aff28 aff28_cneg_lazy(const Aff28& p, bool neg) { return Aff28{p.x, f28_cneg_lazy(p.y, neg)}; }

// 7 multiplications; D = 4 X Y^2 as a product instead of 2((X+B)^2 - A - C)
// (Y3 = E (D - X3) - 8 B^2 as ONE fused difference of products: 6 products' worth of reductions instead of 7, fp28.hpp)
CPX_PT_FN Jac28 jac28_dbl(const Jac28& p) {
  const F28 A = f28_sqr(p.x), B = f28_sqr(p.y);
  const F28 D = f28_shl<2>(f28_mul(p.x, B));
  const F28 E = f28_add(f28_shl<1>(A), A), F = f28_sqr(E);
  Jac28 r;
  r.x = f28_sub_sub2(F, F28::zero(), D);
  r.y = f28_mulsub_body(E, f28_sub_lazy(D, r.x), f28_shl<3>(B), B);
  r.z = f28_shl<1>(f28_mul(p.y, p.z));   // exact zero stays exact zero
  return r;
}

// 11 multiplications; complete
CPX_PT_FN Jac28 jac28_add_mixed(const Jac28& p, const Aff28& q) {
  if (q.is_identity()) return p;
  if (p.is_identity()) return Jac28::from_affine(q);
  const F28 z1z1 = f28_sqr(p.z);
  const F28 u2 = f28_mul(q.x, z1z1);
  const F28 s2 = f28_mul(f28_mul(q.y, p.z), z1z1);
  const F28 h = f28_sub(u2, p.x);
  const F28 zh = f28_mul(p.z, h);
  F28 rr = f28_sub(s2, p.y);
  if (f28_product_is_zero(zh)) {   // H = 0 mod p: P = +-Q
    if (f28_product_is_zero(f28_mul(rr, f28_one()))) return jac28_dbl(p);
    return Jac28::identity();
  }
  rr = f28_shl<1>(rr);
  const F28 i = f28_shl<2>(f28_sqr(h));
  const F28 j = f28_mul(h, i);
  const F28 v = f28_mul(p.x, i);
  Jac28 r;
  r.x = f28_sub(f28_sub(f28_sqr(rr), j), f28_shl<1>(v));
  r.y = f28_sub(f28_mul(rr, f28_sub(v, r.x)), f28_shl<1>(f28_mul(p.y, j)));
  r.z = f28_shl<1>(zh);
  return r;
}

// 16 multiplications; complete
CPX_PT_FN Jac28 jac28_add(const Jac28& p, const Jac28& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  const F28 z1z1 = f28_sqr(p.z), z2z2 = f28_sqr(q.z);
  const F28 u1 = f28_mul(p.x, z2z2), u2 = f28_mul(q.x, z1z1);
  const F28 s1 = f28_mul(f28_mul(p.y, q.z), z2z2), s2 = f28_mul(f28_mul(q.y, p.z), z1z1);
  const F28 h = f28_sub(u2, u1);
  const F28 zzh = f28_mul(f28_mul(p.z, q.z), h);
  F28 rr = f28_sub(s2, s1);
  if (f28_product_is_zero(zzh)) {
    if (f28_product_is_zero(f28_mul(rr, f28_one()))) return jac28_dbl(p);
    return Jac28::identity();
  }
  rr = f28_shl<1>(rr);
  const F28 i = f28_sqr(f28_shl<1>(h));
  const F28 j = f28_mul(h, i);
  const F28 v = f28_mul(u1, i);
  Jac28 r;
  r.x = f28_sub(f28_sub(f28_sqr(rr), j), f28_shl<1>(v));
  r.y = f28_sub(f28_mul(rr, f28_sub(v, r.x)), f28_shl<1>(f28_mul(s1, j)));
  r.z = f28_shl<1>(zzh);
  return r;
}

// ---- extended Jacobian ("XYZZ": x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2) accumulators for the bucket loops ----
// A mixed addition is 8M + 2S (madd-2008-s) instead of the Jacobian 7M + 4S: 3738 instead of 3948 limb products.
// Magnitudes: |X| <= 6.3, |Y| <= 2.7, ZZ and ZZZ are products; the largest operand product is R (<= 4.5) x (Q - X3)
// (<= 8.1).  The zero test for "same x" rides on the product ZZ3 = ZZ1 * PP.
struct Xyzz28 {
  F28 x, y, zz, zzz;
  CPX_HD bool is_identity() const { return zz.is_zero_exact(); }
  static CPX_HD Xyzz28 identity() { return Xyzz28{f28_one(), f28_one(), F28::zero(), F28::zero()}; }
};
CPX_PT_FN Xyzz28 xyzz28_dbl_affine(const Aff28& q) {   // mdbl-2008-s-1: 2 * (x, y)
  const F28 u = f28_shl<1>(q.y), v = f28_sqr(u), w = f28_mul(u, v), s = f28_mul(q.x, v);
  const F28 xx = f28_sqr(q.x), m = f28_add(f28_shl<1>(xx), xx);
  Xyzz28 r;
  r.x = f28_sub(f28_sqr(m), f28_shl<1>(s));
  r.y = f28_sub(f28_mul(m, f28_sub(s, r.x)), f28_mul(w, q.y));
  r.zz = v;
  r.zzz = w;
  return r;
}
// INL: the ten products inlined instead of called (no argument moves: the out-of-line product takes 28 scalar
// arguments and returns 14).  Used by the one loop where it pays, k_msm_tblw: +4 % there, and the kernel drops from
// 256 VGPRs with spills to 231 without; in k_msm_fix (two points in flight) it bought nothing, and inlining
// everywhere would put several 40 KB loops into the instruction cache two CUs share.
template <bool INL> CPX_PT_FN Xyzz28 xyzz28_add_mixed_t(const Xyzz28& p, const Aff28& q) {
  auto mul = [](const F28& a, const F28& b) { return INL ? f28_mul_body(a, b) : f28_mul(a, b); };
  auto sqr = [](const F28& a) { return INL ? f28_sqr_body(a) : f28_sqr(a); };
  if (q.is_identity()) return p;
  if (p.is_identity()) {   // (q.y may come from aff28_cneg_lazy: a stored coordinate gets its carry pass)
    F28 qy = q.y;
    f28_normalize(qy);
    return Xyzz28{q.x, qy, f28_one(), f28_one()};
  }
  const F28 u2 = mul(q.x, p.zz), s2 = mul(q.y, p.zzz);
  const F28 pp_ = f28_sub_lazy(u2, p.x), rr = f28_sub_lazy(s2, p.y);   // P, R and Q - X3 only feed products: no carry pass
  const F28 pp2 = sqr(pp_);
  const F28 zz3 = mul(p.zz, pp2);
  if (f28_product_is_zero(zz3)) {   // same x: P = +-Q
    if (f28_product_is_zero(f28_mul(rr, f28_one()))) return xyzz28_dbl_affine(q);
    return Xyzz28::identity();
  }
  const F28 ppp = mul(pp_, pp2), qq = mul(p.x, pp2);
  Xyzz28 r;
  r.x = f28_sub_sub2(sqr(rr), ppp, qq);
  // the two products of Y3 share one Montgomery reduction where the products are inlined (the bucket loops)
  if (INL) r.y = f28_mulsub_body(rr, f28_sub_lazy(qq, r.x), p.y, ppp);
  else r.y = f28_sub(mul(rr, f28_sub_lazy(qq, r.x)), mul(p.y, ppp));
  r.zz = zz3;
  r.zzz = mul(p.zzz, ppp);
  return r;
}
CPX_PT_FN Xyzz28 xyzz28_add_mixed(const Xyzz28& p, const Aff28& q) { return xyzz28_add_mixed_t<false>(p, q); }
// 2 P for a full XYZZ point (dbl-2008-s-1): 6M + 3S
CPX_PT_FN Xyzz28 xyzz28_dbl(const Xyzz28& p) {
  if (p.is_identity()) return p;
  const F28 u = f28_shl<1>(p.y), v = f28_sqr(u), w = f28_mul(u, v), s = f28_mul(p.x, v);
  const F28 xx = f28_sqr(p.x), m = f28_add(f28_shl<1>(xx), xx);
  Xyzz28 r;
  r.x = f28_sub(f28_sqr(m), f28_shl<1>(s));
  r.y = f28_sub(f28_mul(m, f28_sub(s, r.x)), f28_mul(w, p.y));
  r.zz = f28_mul(v, p.zz);
  r.zzz = f28_mul(w, p.zzz);
  return r;
}
// P + Q for two full XYZZ points (add-2008-s), complete: 12M + 2S, the two products of Y3 under one reduction — 5110
// multiply-adds against the 5817 of the Jacobian addition (11M + 5S).  What the reduction kernels add lane accumulators with.
// Magnitudes (units of p): products lie in (-0.81, 1.81); P, R within +-3.62; X3 in (-6.24, 4.24); Q - X3 within +-8.1;
// Y3 within +-2.62 (+ the slack of the shared reduction): every operand product stays far below 2^11.3 p^2.
CPX_PT_FN Xyzz28 xyzz28_add(const Xyzz28& p, const Xyzz28& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  const F28 u1 = f28_mul(p.x, q.zz), u2 = f28_mul(q.x, p.zz), s1 = f28_mul(p.y, q.zzz), s2 = f28_mul(q.y, p.zzz);
  const F28 pp_ = f28_sub_lazy(u2, u1), rr = f28_sub_lazy(s2, s1);
  const F28 pp2 = f28_sqr(pp_);
  const F28 zz3 = f28_mul(f28_mul(p.zz, q.zz), pp2);
  if (f28_product_is_zero(zz3)) {   // same x: P = +-Q
    if (f28_product_is_zero(f28_mul(rr, f28_one()))) return xyzz28_dbl(p);
    return Xyzz28::identity();
  }
  const F28 ppp = f28_mul(pp_, pp2), qq = f28_mul(u1, pp2);
  Xyzz28 r;
  r.x = f28_sub_sub2(f28_sqr(rr), ppp, qq);
  r.y = f28_mulsub_body(rr, f28_sub_lazy(qq, r.x), s1, ppp);
  r.zz = zz3;
  r.zzz = f28_mul(f28_mul(p.zzz, q.zzz), ppp);
  return r;
}
// same point in Jacobian coordinates with Z = ZZ: (X ZZ, Y ZZZ, ZZ)
CPX_HD Jac28 xyzz28_to_jac(const Xyzz28& p) {
  if (p.is_identity()) return Jac28::identity();
  return Jac28{f28_mul(p.x, p.zz), f28_mul(p.y, p.zzz), p.zz};
}

CPX_HD Aff28 jac28_to_affine_with_zinv(const Jac28& p, const F28& zinv) {
  const F28 zi2 = f28_sqr(zinv);
  return Aff28{f28_mul(p.x, zi2), f28_mul(f28_mul(p.y, zi2), zinv)};
}

// conversions at the edge of the table kernels
CPX_HD Aff28 aff28_from_std(const Aff& a) {
  if (a.is_identity()) return Aff28::identity();
  return Aff28{f28_from_std(a.x), f28_from_std(a.y)};
}
CPX_HD Aff aff28_to_std(const Aff28& a) {
  if (a.is_identity()) return Aff::identity();
  return Aff{f28_to_std(a.x), f28_to_std(a.y)};
}
CPX_HD Jac28 jac28_from_std(const Jac& p) {
  if (p.is_identity()) return Jac28::identity();
  return Jac28{f28_from_std(p.x), f28_from_std(p.y), f28_from_std(p.z)};
}
CPX_HD Jac jac28_to_std(const Jac28& p) {
  if (p.is_identity()) return Jac::identity();
  return Jac{f28_to_std(p.x), f28_to_std(p.y), f28_to_std(p.z)};
}

// Fermat inverse in the lazy Montgomery domain
CPX_HD F28 f28_inv(const F28& a) {
  F28 r = f28_one();
  bool started = false;
  for (int l = 11; l >= 0; l--) {
    uint32_t w = FpCfg::P[l];
    if (l == 0) w -= 2;
    for (int b = 31; b >= 0; b--) {
      if (started) r = f28_sqr(r);
      if ((w >> b) & 1u) {
        r = started ? f28_mul(r, a) : a;
        started = true;
      }
    }
  }
  return r;
}


// a^((p+1)/4): the square root of a if a is a quadratic residue (p = 3 mod 4).  Sliding window of width 3 over the constant exponent
// (odd powers a, a^3, a^5, a^7 in registers): 378 squarings + 97 products + 4 for the table, where plain square-and-multiply paid one
// product per set bit (190).  The exponent is a constant, so every branch below is uniform over a wave.
CPX_HD F28 f28_sqrt_candidate(const F28& a) {
  // (p+1)/4, little-endian 32-bit words
  const uint32_t E[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                          0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
  auto bit = [&](int i) { return (E[i >> 5] >> (i & 31)) & 1u; };
  const F28 a2 = f28_sqr(a), a3 = f28_mul(a2, a), a5 = f28_mul(a3, a2), a7 = f28_mul(a5, a2);
  F28 r = f28_one();
  bool started = false;
  int i = 383;
  while (i >= 0) {
    if (!bit(i)) {
      if (started) r = f28_sqr(r);
      i--;
      continue;
    }
    int j = i - 2 < 0 ? 0 : i - 2;   // the window [i .. j] ends in a set bit
    while (!bit(j)) j++;
    uint32_t v = 0;
    for (int t = i; t >= j; t--) v = (v << 1) | bit(t);
    // (one call per table entry instead of a reference picked among the four: hipcc turned `const F28& f = v == 1 ? a : ...` into a
    // select between flat pointers to spilled limbs, and the device function then read garbage — the CPU build was fine)
    if (started) {
      for (int t = i; t >= j; t--) r = f28_sqr(r);
      if (v == 1) r = f28_mul(r, a);
      else if (v == 3) r = f28_mul(r, a3);
      else if (v == 5) r = f28_mul(r, a5);
      else r = f28_mul(r, a7);
    } else {
      if (v == 1) r = a;
      else if (v == 3) r = a3;
      else if (v == 5) r = a5;
      else r = a7;
      started = true;
    }
    i = j - 1;
  }
  return r;
}
// a == b mod p for lazy values
CPX_HD bool f28_eq(const F28& a, const F28& b) { return f28_product_is_zero(f28_mul(f28_sub(a, b), f28_one())); }

// Membership of an on-curve affine point in the order-r subgroup by the endomorphism test (M. Scott, "A note on
// group membership tests for G1, G2 and GT on BLS pairing-friendly curves", 2021): phi(x, y) = (beta x, y) satisfies
// phi^2 + phi + 1 = 0 on the whole curve and acts on G1 as [-u^2], and u^4 - u^2 + 1 = r exactly, hence
// phi(P) = [-u^2] P  <=>  [r] P = O.  Same verdict as ark-ec's [r]P == O (reference: deserialize_compressed with
// Validate::Yes, whisk.rs:265-277), at two 64-bit multiplications by |u| = 0xd201000000010000 instead of a 255-bit one.
CPX_HD bool g1_28_in_subgroup(const Aff28& P) {
  if (P.is_identity()) return true;
  const uint64_t U = 0xd201000000010000ull;
  Jac28 q = Jac28::from_affine(P);             // [|u|] P
  for (int b = 62; b >= 0; b--) {
    q = jac28_dbl(q);
    if ((U >> b) & 1) q = jac28_add_mixed(q, P);
  }
  const Jac28 q1 = q;                          // [u^2] P = [|u|] ([|u|] P)
  for (int b = 62; b >= 0; b--) {
    q = jac28_dbl(q);
    if ((U >> b) & 1) q = jac28_add(q, q1);
  }
  if (q.is_identity()) return false;
  // (beta x_P, y_P) == (X / Z^2, -Y / Z^3)
  const F28 zz = f28_sqr(q.z);
  const F28 lhs_x = f28_mul(f28_mul(f28_const(F28Cfg::BETA), P.x), zz);
  const F28 lhs_y = f28_mul(f28_mul(P.y, zz), q.z);
  return f28_eq(lhs_x, q.x) && f28_eq(lhs_y, f28_neg(q.y));
}

}  // namespace cpx

// ------------------------------------------------------------------ kernel-facing names
// Tables, partial sums and every group operation of the MSM / build / finalise kernels live in this carry-free
// 28-bit-limb "table form"; everything that crosses the C-ABI, the instance vectors and the proof slots stay in the
// standard 12 x u32 Montgomery form (mont32.hpp, g1.hpp).  Conversion happens where points enter a kernel (t_from_std)
// and where results leave it (t_to_std).  The t_* names keep the kernels independent of the representation.
namespace cpx {

typedef F28 TF;
typedef Aff28 TAff;
// An entry of the fixed-base table of multiples (k_msm_fix gathers them at random from a 15 GB table): padded to one
// 128-byte line, so that a gather touches ONE cache line / HBM burst instead of straddling two (112-byte entries at
// 112-byte strides cross a 128-byte boundary 7 times out of 8).  -DCPX_FIXTAB_PAD=0 selects the dense layout (A/B runs).
#ifndef CPX_FIXTAB_PAD
#define CPX_FIXTAB_PAD 1
#endif
#if CPX_FIXTAB_PAD
struct alignas(128) TFix {
  Aff28 a;
};
#else
struct TFix {
  Aff28 a;
};
#endif
typedef Jac28 TJac;
CPX_HD TF t_one() { return f28_one(); }
CPX_HD TF t_mul(const TF& a, const TF& b) { return f28_mul(a, b); }
CPX_HD bool t_is_zero_raw(const TF& a) { return a.is_zero_exact(); }
CPX_HD TJac t_dbl(const TJac& p) { return jac28_dbl(p); }
CPX_HD TJac t_add_mixed(const TJac& p, const TAff& q) { return jac28_add_mixed(p, q); }
CPX_HD TJac t_add(const TJac& p, const TJac& q) { return jac28_add(p, q); }
CPX_HD TAff t_cneg(const TAff& p, bool neg) { return aff28_cneg(p, neg); }
CPX_HD TAff t_cneg_lazy(const TAff& p, bool neg) { return aff28_cneg_lazy(p, neg); }
CPX_HD TAff t_to_affine(const TJac& p, const TF& zinv) { return jac28_to_affine_with_zinv(p, zinv); }
CPX_HD TAff t_from_std(const Aff& a) { return aff28_from_std(a); }
CPX_HD Aff t_to_std(const TAff& a) { return aff28_to_std(a); }
CPX_HD Jac t_jac_to_std(const TJac& p) { return jac28_to_std(p); }
CPX_HD TJac t_jac_from_std(const Jac& p) { return jac28_from_std(p); }
// bucket accumulators: extended Jacobian coordinates while points are being added, Jacobian for the reductions
typedef Xyzz28 TAcc;
CPX_HD TAcc t_acc_add_mixed(const TAcc& p, const TAff& q) { return xyzz28_add_mixed(p, q); }
CPX_HD TAcc t_acc_add_mixed_inl(const TAcc& p, const TAff& q) { return xyzz28_add_mixed_t<true>(p, q); }
CPX_HD TJac t_acc_to_jac(const TAcc& p) { return xyzz28_to_jac(p); }
CPX_HD TAcc t_acc_add(const TAcc& p, const TAcc& q) { return xyzz28_add(p, q); }
CPX_HD TAcc t_acc_dbl(const TAcc& p) { return xyzz28_dbl(p); }
CPX_HD TF t_inv(const TF& a) { return f28_inv_euclid(a); }
CPX_HD TF t_neg(const TF& a) { return f28_neg(a); }
CPX_HD TF t_beta() { return f28_const(F28Cfg::BETA); }   // (x, y) -> (beta x, y) is multiplication by -z^2 on G1 (glv.hpp)

}  // namespace cpx
