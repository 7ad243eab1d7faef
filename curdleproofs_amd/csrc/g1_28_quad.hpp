// Point additions of the latency paths, spread over the four lanes of a quad — device only.
//
// A lone proof (or a few) is a chain of DEPENDENT point additions: the trees of k_reduce_sets_wave / k_finalize_ranges_wave / the
// MSM tails keep 1-32 lanes of a wave busy, and a lane needs 16-20 us per addition because the 14-16 field products of one addition
// run one after the other (1.03 us each, profiles/r02_single_proof_latency.txt).  The products of an extended-Jacobian (XYZZ)
// addition form a dependency graph of depth 4 and width <= 4:
//     round 1   U1 = X1 ZZ2     U2 = X2 ZZ1      S1 = Y1 ZZZ2      S2 = Y2 ZZZ1          P = U2 - U1, R = S2 - S1
//     round 2   PP = P^2        RR = R^2         ZZ12 = ZZ1 ZZ2    ZZZ12 = ZZZ1 ZZZ2
//     round 3   PPP = P PP      Q = U1 PP        ZZ3 = ZZ12 PP     (R 1: zero test)      X3 = RR - PPP - 2 Q
//     round 4   R (Q - X3)      S1 PPP           ZZZ3 = ZZZ12 PPP                         Y3 = R (Q - X3) - S1 PPP
// so four lanes finish it in the time of 4 products instead of 14.  All four lanes of a quad hold the SAME operands and return the
// SAME result (replicated): every lane multiplies the operand pair its sub-lane index selects and the products travel with DPP
// quad_perm broadcasts (v_mov_b32_dpp, no LDS, no waiting).  The rare cases (an identity operand, P = +-Q) are decided on replicated
// data, i.e. identically in the four lanes, and the doubling is then done by every lane on its own from the operands in memory.
// Magnitudes (units of p): products lie in (-0.81, 1.81); P, R within +-3.62; X3 in (-6.24, 4.24); Q - X3 within +-8.1;
// Y3 within +-2.62 — every operand product stays far below the 2^11.3 p^2 the lazy multiplication allows (fp28.hpp).
// Every function here must be called by all 64 lanes of the wave together (DPP reads from disabled lanes are undefined).
#pragma once
#include "g1_28.hpp"

#if defined(__HIPCC__)
namespace cpx {

template <int K> __device__ __forceinline__ F28 quad_bcast(const F28& a) {   // the value lane K of each quad holds
  F28 r;
  CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = __builtin_amdgcn_mov_dpp(a.v[i], K * 0x55, 0xf, 0xf, false);
  return r;
}
__device__ __forceinline__ F28 quad_sel(int sub, const F28& a0, const F28& a1, const F28& a2, const F28& a3) {
  F28 r;
  CPX_UNROLL for (int i = 0; i < 14; i++) {
    const int32_t lo = (sub & 1) ? a1.v[i] : a0.v[i], hi = (sub & 1) ? a3.v[i] : a2.v[i];
    r.v[i] = (sub & 2) ? hi : lo;
  }
  return r;
}

// P + Q.  Operands and result replicated over the quad.  Returns QUAD_OK with r = P + Q in the generic case; otherwise r is
// meaningless and the code tells the caller — who still has the operands in memory, here they would pin 112 registers to the end —
// what the sum is (identical in the four lanes).
enum : int { QUAD_OK = 0, QUAD_TAKE_Q = 1, QUAD_TAKE_P = 2, QUAD_DOUBLE_P = 3, QUAD_IDENTITY = 4 };
__device__ __forceinline__ int xyzz28_add_quad_core(const Xyzz28& p, const Xyzz28& q, Xyzz28& r) {
  const int sub = threadIdx.x & 3;
  const int idcode = p.is_identity() ? QUAD_TAKE_Q : q.is_identity() ? QUAD_TAKE_P : QUAD_OK;
  F28 t = f28_mul(quad_sel(sub, p.x, q.x, p.y, q.y), quad_sel(sub, q.zz, p.zz, q.zzz, p.zzz));
  const F28 u1 = quad_bcast<0>(t), s1 = quad_bcast<2>(t);
  const F28 pp_ = f28_sub_lazy(quad_bcast<1>(t), u1), rr = f28_sub_lazy(quad_bcast<3>(t), s1);   // feed products only: no carry pass
  t = f28_mul(quad_sel(sub, pp_, rr, p.zz, p.zzz), quad_sel(sub, pp_, rr, q.zz, q.zzz));
  const F28 pp2 = quad_bcast<0>(t), rr2 = quad_bcast<1>(t), zz12 = quad_bcast<2>(t), zzz12 = quad_bcast<3>(t);
  t = f28_mul(quad_sel(sub, pp_, u1, zz12, rr), quad_sel(sub, pp2, pp2, pp2, f28_one()));
  const F28 ppp = quad_bcast<0>(t), qq = quad_bcast<1>(t), r1 = quad_bcast<3>(t);
  r.zz = quad_bcast<2>(t);
  r.x = f28_sub_sub2(rr2, ppp, qq);
  t = f28_mul(quad_sel(sub, rr, s1, zzz12, zzz12), quad_sel(sub, f28_sub_lazy(qq, r.x), ppp, ppp, ppp));
  r.y = f28_sub(quad_bcast<0>(t), quad_bcast<1>(t));
  r.zzz = quad_bcast<2>(t);
  if (idcode != QUAD_OK) return idcode;
  if (f28_product_is_zero(r.zz)) return f28_product_is_zero(r1) ? QUAD_DOUBLE_P : QUAD_IDENTITY;   // same x: P = +-Q
  return QUAD_OK;
}
// buf[dst] = buf[ia] + buf[ib] for operands in (shared) memory: the quad loads both, adds, lane `sub` 0 stores; complete.
__device__ __forceinline__ Xyzz28 xyzz28_add_quad_mem(const Xyzz28* buf, int ia, int ib) {
  Xyzz28 r;
  const int code = xyzz28_add_quad_core(buf[ia], buf[ib], r);
  if (code == QUAD_OK) return r;
  if (code == QUAD_TAKE_Q) return buf[ib];
  if (code == QUAD_TAKE_P) return buf[ia];
  if (code == QUAD_DOUBLE_P) return xyzz28_dbl(buf[ia]);
  return Xyzz28::identity();
}

// 2 P over a quad: V = U^2, XX = X^2 | W = U V, S = X V, MM = M^2 | M (S - X3), W Y, V ZZ, W ZZZ  (3 product rounds instead of 9)
__device__ __forceinline__ Xyzz28 xyzz28_dbl_quad(const Xyzz28& p) {
  const int sub = threadIdx.x & 3;
  const F28 u = f28_shl<1>(p.y);
  F28 t = f28_mul(quad_sel(sub, u, p.x, u, p.x), quad_sel(sub, u, p.x, u, p.x));
  const F28 v = quad_bcast<0>(t), xx = quad_bcast<1>(t);
  const F28 m = f28_add(f28_shl<1>(xx), xx);
  t = f28_mul(quad_sel(sub, u, p.x, m, m), quad_sel(sub, v, v, m, m));
  const F28 w = quad_bcast<0>(t), s = quad_bcast<1>(t), mm = quad_bcast<2>(t);
  Xyzz28 r;
  r.x = f28_sub_sub2(mm, F28::zero(), s);
  t = f28_mul(quad_sel(sub, m, w, v, w), quad_sel(sub, f28_sub_lazy(s, r.x), p.y, p.zz, p.zzz));
  r.y = f28_sub(quad_bcast<0>(t), quad_bcast<1>(t));
  r.zz = quad_bcast<2>(t);
  r.zzz = quad_bcast<3>(t);
  if (p.is_identity()) return p;
  return r;
}

// 2 P in Jacobian coordinates over a quad (the formulas of jac28_dbl): A = X^2, B = Y^2, Y Z | C = B^2, X B, F = (3A)^2 | E (D - X3):
// the chain Y -> Y^2 -> (B^2, X B) -> E (D - X3) is three products deep, so three rounds per doubling instead of seven products.
__device__ __forceinline__ Jac28 jac28_dbl_quad(const Jac28& p) {
  const int sub = threadIdx.x & 3;
  F28 t = f28_mul(quad_sel(sub, p.x, p.y, p.y, p.y), quad_sel(sub, p.x, p.y, p.z, p.z));
  const F28 A = quad_bcast<0>(t), B = quad_bcast<1>(t), yz = quad_bcast<2>(t);
  const F28 E = f28_add(f28_shl<1>(A), A);
  t = f28_mul(quad_sel(sub, B, p.x, E, E), quad_sel(sub, B, B, E, E));
  const F28 C = quad_bcast<0>(t), D = f28_shl<2>(quad_bcast<1>(t)), F = quad_bcast<2>(t);
  Jac28 r;
  r.x = f28_sub(F, f28_shl<1>(D));
  r.y = f28_sub(f28_mul(E, f28_sub_lazy(D, r.x)), f28_shl<3>(C));   // the same product in the four lanes: nothing to exchange
  r.z = f28_shl<1>(yz);                                        // exact zero stays exact zero
  return r;
}

// Jacobian (X, Y, Z) <-> XYZZ (X, Y, Z^2, Z^3): the partial-sum arrays between the kernels stay Jacobian
__device__ __forceinline__ Xyzz28 xyzz28_from_jac(const Jac28& p) {
  if (p.is_identity()) return Xyzz28::identity();
  const F28 zz = f28_sqr(p.z);
  return Xyzz28{p.x, p.y, zz, f28_mul(zz, p.z)};
}

}  // namespace cpx
#endif
