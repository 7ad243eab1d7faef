// Arithmetic of the TABLE kernels (k_table_build, k_fix_build, k_msm_tblw, k_msm_fix, k_finalize_ranges).
// Tables and the partial sums of those kernels live in the carry-free 28-bit-limb representation (fp28.cuh,
// g1_28.cuh); everything that crosses the C-ABI, the instance vectors, the proof slots and the non-table
// kernels stay in the standard 12 x u32 Montgomery form.  Conversion happens where a table is built (standard
// source points in) and in k_finalize_ranges (standard affine + compressed bytes out).
// -DCPX_F28=0 builds the same kernels on the 32-bit-limb field (A/B measurements).
#pragma once
#include "g1.cuh"
#include "g1_28.cuh"

#ifndef CPX_F28
#define CPX_F28 1
#endif

namespace cpx {

#if CPX_F28
typedef F28 TF;
typedef Aff28 TAff;
typedef Jac28 TJac;
CPX_HD TF t_one() { return f28_one(); }
CPX_HD TF t_mul(const TF& a, const TF& b) { return f28_mul(a, b); }
CPX_HD bool t_is_zero_raw(const TF& a) { return a.is_zero_exact(); }
CPX_HD TJac t_dbl(const TJac& p) { return jac28_dbl(p); }
CPX_HD TJac t_add_mixed(const TJac& p, const TAff& q) { return jac28_add_mixed(p, q); }
CPX_HD TJac t_add(const TJac& p, const TJac& q) { return jac28_add(p, q); }
CPX_HD TJac t_dbl_fn(const TJac& p) { return jac28_dbl_fn(p); }
CPX_HD TJac t_add_mixed_fn(const TJac& p, const TAff& q) { return jac28_add_mixed_fn(p, q); }
CPX_HD TJac t_add_fn(const TJac& p, const TJac& q) { return jac28_add_fn(p, q); }
CPX_HD TAff t_cneg(const TAff& p, bool neg) { return aff28_cneg(p, neg); }
CPX_HD TAff t_to_affine(const TJac& p, const TF& zinv) { return jac28_to_affine_with_zinv(p, zinv); }
CPX_HD TAff t_from_std(const Aff& a) { return aff28_from_std(a); }
CPX_HD Aff t_to_std(const TAff& a) { return aff28_to_std(a); }
CPX_HD Jac t_jac_to_std(const TJac& p) { return jac28_to_std(p); }
CPX_HD TJac t_jac_from_std(const Jac& p) { return jac28_from_std(p); }
CPX_HD TF t_inv(const TF& a) { return f28_inv_euclid(a); }
#else
typedef Fp TF;
typedef Aff TAff;
typedef Jac TJac;
CPX_HD TF t_one() { return Fp::one(); }
CPX_HD TF t_mul(const TF& a, const TF& b) { return fe_mul(a, b); }
CPX_HD bool t_is_zero_raw(const TF& a) { return a.is_zero(); }
CPX_HD TJac t_dbl(const TJac& p) { return jac_dbl(p); }
CPX_HD TJac t_add_mixed(const TJac& p, const TAff& q) { return jac_add_mixed(p, q); }
CPX_HD TJac t_add(const TJac& p, const TJac& q) { return jac_add(p, q); }
CPX_HD TJac t_dbl_fn(const TJac& p) { return jac_dbl_fn(p); }
CPX_HD TJac t_add_mixed_fn(const TJac& p, const TAff& q) { return jac_add_mixed_fn(p, q); }
CPX_HD TJac t_add_fn(const TJac& p, const TJac& q) { return jac_add_fn(p, q); }
CPX_HD TAff t_cneg(const TAff& p, bool neg) { return aff_cneg(p, neg); }
CPX_HD TAff t_to_affine(const TJac& p, const TF& zinv) { return jac_to_affine_with_zinv(p, zinv); }
CPX_HD TAff t_from_std(const Aff& a) { return a; }
CPX_HD Aff t_to_std(const TAff& a) { return a; }
CPX_HD Jac t_jac_to_std(const TJac& p) { return p; }
CPX_HD TJac t_jac_from_std(const Jac& p) { return p; }
CPX_HD TF t_inv(const TF& a) { return fe_inv_euclid(a); }
#endif

}  // namespace cpx
