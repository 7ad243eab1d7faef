//! Replacement bodies for `src/util.rs` under `--features mi355x` (reference util.rs:19-29, :83-106) and the two group-side
//! loops that bypass `util::msm`.  NOT COMPILED in the build image.
#![allow(non_snake_case)]

use ark_bls12_381::{Fr, G1Affine, G1Projective};
use ark_std::rand::RngCore;

use crate::crs::CurdleproofsCrs;
use crate::ffi::*;
use crate::util::{generate_blinders, get_permutation};
use crate::N_BLINDERS;

/// util.rs:19-22
pub fn msm(points: &[G1Affine], scalars: &[Fr]) -> G1Projective {
    assert_eq!(points.len(), scalars.len());
    let wire = affine_to_wire(points);
    let mut out = [0u8; JAC];
    let rc = unsafe { cpx_g1_msm(ctx(), wire.as_ptr(), scalars_ptr(scalars), points.len(), out.as_mut_ptr()) };
    assert_eq!(rc, CPX_OK, "cpx_g1_msm");
    projective_from_wire(&out) // some representative of the class; callers compare / normalise
}

/// util.rs:25-29 (the batch normalisation happens on the device)
pub fn msm_from_projective(points: &[G1Projective], scalars: &[Fr]) -> G1Projective {
    assert_eq!(points.len(), scalars.len());
    let mut out = [0u8; JAC];
    let rc = unsafe { cpx_g1_msm_jac(ctx(), projective_to_wire(points).as_ptr(), scalars_ptr(scalars), points.len(), out.as_mut_ptr()) };
    assert_eq!(rc, CPX_OK, "cpx_g1_msm_jac");
    projective_from_wire(&out)
}

/// inner_product_argument.rs:177-178 / same_multiscalar_argument.rs:128-130:
/// `for i in 0..n { vec_L[i] = (vec_L[i] + vec_R[i] * gamma).into_affine() }`
pub fn fold_bases(vec_L: &mut [G1Affine], vec_R: &[G1Affine], gamma: &Fr) {
    assert_eq!(vec_L.len(), vec_R.len());
    let mut l = affine_to_wire(vec_L);
    let r = affine_to_wire(vec_R);
    let rc = unsafe { cpx_g1_fold(ctx(), l.as_mut_ptr(), r.as_ptr(), scalars_ptr(std::slice::from_ref(gamma)), vec_L.len()) };
    assert_eq!(rc, CPX_OK, "cpx_g1_fold");
    vec_L.copy_from_slice(&affine_from_wire(&l));
}

/// grand_product_argument.rs:90-102 (one scalar per point) and util.rs:94-95 (one shared scalar):
/// `points[i].mul(scalars[i]).into_affine()`
pub fn scale_points(points: &[G1Affine], scalars: &[Fr]) -> Vec<G1Affine> {
    assert!(scalars.len() == points.len() || scalars.len() == 1);
    let p = affine_to_wire(points);
    let mut out = vec![0u8; AFF * points.len()];
    let stride = if scalars.len() == 1 { 0 } else { FR };
    let rc = unsafe { cpx_g1_scale(ctx(), p.as_ptr(), scalars_ptr(scalars), stride, points.len(), out.as_mut_ptr()) };
    assert_eq!(rc, CPX_OK, "cpx_g1_scale");
    affine_from_wire(&out)
}

/// util.rs:83-106
pub fn shuffle_permute_and_commit_input<T: RngCore>(
    crs: &CurdleproofsCrs,
    vec_R: &[G1Affine],
    vec_S: &[G1Affine],
    permutation: &[u32],
    k: &Fr,
    rng: &mut T,
) -> (Vec<G1Affine>, Vec<G1Affine>, G1Projective, Vec<Fr>) {
    let ell = crs.vec_G.len();
    let kk = [*k];
    let vec_T = get_permutation(&scale_points(vec_R, &kk), permutation);
    let vec_U = get_permutation(&scale_points(vec_S, &kk), permutation);
    let range_as_fr: Vec<Fr> = (0..ell as u32).map(Fr::from).collect();
    let sigma_ell = get_permutation(&range_as_fr, permutation);
    let vec_m_blinders = generate_blinders(rng, N_BLINDERS); // the same single draw site as the reference (util.rs:91)
    let M = msm(&crs.vec_G, &sigma_ell) + msm(&crs.vec_H, &vec_m_blinders);
    (vec_T, vec_U, M, vec_m_blinders)
}
