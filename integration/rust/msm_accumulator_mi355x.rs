//! `MsmAccumulator` over `cpx_accum_*` (reference msm_accumulator.rs:22-68) under `--features mi355x`.  NOT COMPILED here.
#![allow(non_snake_case)]

use ark_bls12_381::{Fr, G1Affine, G1Projective};
use ark_std::rand::RngCore;
use ark_std::UniformRand;

use crate::errors::ProofError;
use crate::ffi::*;

pub struct MsmAccumulator {
    h: *mut cpx_accum,
}

impl MsmAccumulator {
    pub(crate) fn new() -> Self {
        let mut h: *mut cpx_accum = std::ptr::null_mut();
        let rc = unsafe { cpx_accum_new(ctx(), &mut h) };
        assert_eq!(rc, CPX_OK, "cpx_accum_new");
        Self { h }
    }

    /// msm_accumulator.rs:38-52.  The factor is drawn here, from the caller's rng, exactly where the reference draws it (:44);
    /// the library refuses a zero factor (it would drop the check), which `Fr::rand` returns with probability 1/r: redraw.
    pub fn accumulate_check<T: RngCore>(&mut self, C: &G1Projective, vec_x: &[Fr], vec_V: &[G1Affine], rng: &mut T) {
        let mut random_factor = Fr::rand(rng);
        while random_factor == Fr::from(0u64) {
            random_factor = Fr::rand(rng);
        }
        let n = vec_x.len().min(vec_V.len()); // `zip`
        let v = affine_to_wire(&vec_V[..n]);
        let c_wire = projective_to_wire(std::slice::from_ref(C)); // field by field: Projective is repr(Rust)
        let rc = unsafe {
            cpx_accum_check(self.h, c_wire.as_ptr(), scalars_ptr(vec_x), v.as_ptr(), n, scalars_ptr(std::slice::from_ref(&random_factor)))
        };
        assert_eq!(rc, CPX_OK, "cpx_accum_check");
    }

    /// msm_accumulator.rs:55-68
    pub fn verify(self) -> Result<(), ProofError> {
        match unsafe { cpx_accum_verify(self.h) } {
            CPX_OK => Ok(()),
            CPX_ERR_VERIFY => Err(ProofError::VerificationError),
            rc => panic!("cpx_accum_verify: {}", rc),
        }
    }
}

impl Drop for MsmAccumulator {
    fn drop(&mut self) {
        unsafe { cpx_accum_free(self.h) }
    }
}
