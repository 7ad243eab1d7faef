//! `CurdleproofsProof::{new, verify}` over tier 2 of the C-ABI (reference curdleproofs.rs:59-195, :197-298) and the batch
//! forms the benchmark measures.  Goes into `impl CurdleproofsProof` under `--features mi355x`.  NOT COMPILED here.
#![allow(non_snake_case)]

use ark_bls12_381::{Fr, G1Affine, G1Projective};
use ark_std::rand::RngCore;
use ark_std::UniformRand;

use crate::crs::CurdleproofsCrs;
use crate::curdleproofs::CurdleproofsProof;
use crate::errors::ProofError;
use crate::ffi::*;
use crate::N_BLINDERS;

/// The prover's `Fr::rand(rng)` draws in the reference's order: vec_a_blinders[2] (curdleproofs.rs:86),
/// vec_c_blinders[4] (grand_product_argument.rs:75), IPA r[n], z[n-2] (inner_product_argument.rs:46-47), r_t, r_u
/// (curdleproofs.rs:110-111), r_a, r_b, r_k (same_scalar_argument.rs:56-58), vec_r[n] (same_multiscalar_argument.rs:78).
fn prover_draws<T: RngCore>(n: usize, rng: &mut T) -> Vec<Fr> {
    (0..3 * n + 9).map(|_| Fr::rand(rng)).collect()
}
/// The verifier's eight `accumulate_check` factors (msm_accumulator.rs:44) in call order.
fn verifier_draws<T: RngCore>(count: usize, rng: &mut T) -> Vec<Fr> {
    let mut v = Vec::with_capacity(count);
    while v.len() < count {
        let a = Fr::rand(rng);
        if a != Fr::from(0u64) {
            v.push(a); // a zero factor (probability 1/r) would drop a check: the library refuses it
        }
    }
    v
}

impl CurdleproofsProof {
    #[allow(clippy::too_many_arguments)]
    pub fn new<T: RngCore>(
        crs: &CurdleproofsCrs,
        vec_R: Vec<G1Affine>,
        vec_S: Vec<G1Affine>,
        vec_T: Vec<G1Affine>,
        vec_U: Vec<G1Affine>,
        M: G1Projective,
        permutation: Vec<u32>,
        k: Fr,
        vec_m_blinders: Vec<Fr>,
        rng: &mut T,
    ) -> CurdleproofsProof {
        let proofs = Self::new_batch(crs, &[(vec_R, vec_S, vec_T, vec_U, M, permutation, k, vec_m_blinders)], rng);
        proofs.into_iter().next().unwrap()
    }

    /// BASELINE configs[3]: many independent instances of one CRS in one call (instances resident in HBM, one pass of kernels).
    #[allow(clippy::type_complexity)]
    pub fn new_batch<T: RngCore>(
        crs: &CurdleproofsCrs,
        instances: &[(Vec<G1Affine>, Vec<G1Affine>, Vec<G1Affine>, Vec<G1Affine>, G1Projective, Vec<u32>, Fr, Vec<Fr>)],
        rng: &mut T,
    ) -> Vec<CurdleproofsProof> {
        let h = ctx_with_crs(crs);
        let ell = crs.vec_G.len();
        let n = ell + N_BLINDERS;
        let b = instances.len();
        let (mut r, mut s, mut t, mut u) = (Vec::new(), Vec::new(), Vec::new(), Vec::new());
        let (mut m, mut perm, mut ks, mut mbl, mut rand) = (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
        for (vr, vs, vt, vu, mm, p, k, bl) in instances {
            assert!(vr.len() == ell && vs.len() == ell && vt.len() == ell && vu.len() == ell && p.len() == ell && bl.len() == N_BLINDERS);
            r.extend(affine_to_wire(vr));
            s.extend(affine_to_wire(vs));
            t.extend(affine_to_wire(vt));
            u.extend(affine_to_wire(vu));
            m.push(*mm);
            perm.extend_from_slice(p);
            ks.push(*k);
            mbl.extend_from_slice(bl);
            rand.extend(prover_draws(n, rng)); // instance by instance, each in the reference's order
        }
        let psz = unsafe {
            let rc = cpx_batch_load(h, b, r.as_ptr(), s.as_ptr(), t.as_ptr(), u.as_ptr(), projective_to_wire(&m).as_ptr());
            assert_eq!(rc, CPX_OK, "cpx_batch_load");
            cpx_proof_size(h)
        };
        let mut bytes = vec![0u8; b * psz];
        let rc = unsafe { cpx_batch_prove(h, perm.as_ptr(), scalars_ptr(&ks), scalars_ptr(&mbl), scalars_ptr(&rand), bytes.as_mut_ptr()) };
        assert_eq!(rc, CPX_OK, "cpx_batch_prove");
        let log2_n = n.trailing_zeros() as usize;
        bytes.chunks(psz).map(|c| CurdleproofsProof::deserialize(c, log2_n).expect("the library emits canonical encodings")).collect()
    }

    #[allow(clippy::too_many_arguments)]
    pub fn verify<T: RngCore>(
        &self,
        crs: &CurdleproofsCrs,
        vec_R: &Vec<G1Affine>,
        vec_S: &Vec<G1Affine>,
        vec_T: &Vec<G1Affine>,
        vec_U: &Vec<G1Affine>,
        M: &G1Projective,
        rng: &mut T,
    ) -> Result<(), ProofError> {
        let h = ctx_with_crs(crs);
        let mut bytes = Vec::new();
        self.serialize(&mut bytes).map_err(|_| ProofError::VerificationError)?; // curdleproofs.rs:300-310
        let factors = verifier_draws(8, rng);
        let mut verdict: std::os::raw::c_int = CPX_ERR_INTERNAL;
        unsafe {
            let rc = cpx_batch_load(h, 1, affine_to_wire(vec_R).as_ptr(), affine_to_wire(vec_S).as_ptr(), affine_to_wire(vec_T).as_ptr(),
                                    affine_to_wire(vec_U).as_ptr(), projective_to_wire(std::slice::from_ref(M)).as_ptr());
            assert_eq!(rc, CPX_OK, "cpx_batch_load");
            let rc = cpx_batch_verify(h, bytes.as_ptr(), scalars_ptr(&factors), &mut verdict);
            assert_eq!(rc, CPX_OK, "cpx_batch_verify");
        }
        match verdict {
            CPX_OK => Ok(()),
            _ => Err(ProofError::VerificationError), // CPX_ERR_VERIFY; CPX_ERR_DESERIALIZE cannot occur for a proof object
        }
    }

    /// BASELINE configs[4]: ONE accumulated check over all proofs loaded with `cpx_batch_load` on this GPU (12 factors per
    /// proof); returns this GPU's partial sum and the number of undecodable proofs.  The caller all-gathers the 144-byte
    /// partials of all GPUs and accepts iff `cpx_g1_sum_jac` reports the identity and no GPU saw an invalid proof.
    pub fn verify_fused_partial<T: RngCore>(serialized_proofs: &[u8], batch: usize, rng: &mut T) -> ([u8; JAC], usize) {
        let factors = verifier_draws(12 * batch, rng);
        let mut part = [0u8; JAC];
        let mut bad: std::os::raw::c_int = 0;
        let rc = unsafe { cpx_batch_verify_fused(ctx(), serialized_proofs.as_ptr(), scalars_ptr(&factors), part.as_mut_ptr(), &mut bad) };
        assert_eq!(rc, CPX_OK, "cpx_batch_verify_fused");
        (part, bad as usize)
    }
}
