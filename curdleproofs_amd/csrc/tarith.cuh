// Arithmetic of the MSM / build / finalise kernels.  Tables, partial sums and every group operation on the device
// live in the carry-free 28-bit-limb "table form" (fp28.cuh, g1_28.cuh); everything that crosses the C-ABI, the
// instance vectors and the proof slots stay in the standard 12 x u32 Montgomery form (mont32.cuh, g1.cuh).
// Conversion happens where points enter a kernel (t_from_std) and where results leave it (t_to_std).
// The t_* names keep the kernels independent of the representation.
#pragma once
#include "g1.cuh"
#include "g1_28.cuh"

namespace cpx {

typedef F28 TF;
typedef Aff28 TAff;
typedef Jac28 TJac;
CPX_HD TF t_one() { return f28_one(); }
CPX_HD TF t_mul(const TF& a, const TF& b) { return f28_mul(a, b); }
CPX_HD bool t_is_zero_raw(const TF& a) { return a.is_zero_exact(); }
CPX_HD TJac t_dbl(const TJac& p) { return jac28_dbl(p); }
CPX_HD TJac t_add_mixed(const TJac& p, const TAff& q) { return jac28_add_mixed(p, q); }
CPX_HD TJac t_add(const TJac& p, const TJac& q) { return jac28_add(p, q); }
CPX_HD TAff t_cneg(const TAff& p, bool neg) { return aff28_cneg(p, neg); }
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
CPX_HD TF t_inv(const TF& a) { return f28_inv_euclid(a); }
CPX_HD TF t_neg(const TF& a) { return f28_neg(a); }
CPX_HD TF t_beta() { return f28_const(F28Cfg::BETA); }   // (x, y) -> (beta x, y) is multiplication by -z^2 on G1 (glv.cuh)

}  // namespace cpx
