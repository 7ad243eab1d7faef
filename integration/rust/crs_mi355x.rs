//! The hash-to-curve CRS ceremony of the reference's integration test (tests/crs.rs:13-52 `generate_random_points`) with the square
//! roots, the cofactor multiplication and the subgroup test on the MI355X (`cpx_g1_decompress_status`, `cpx_g1_scale`,
//! `cpx_g1_normalize`).  Goes next to the test (or into `src/crs.rs` under `--features mi355x`).  NOT COMPILED here; the same call
//! sequence runs in `curdleproofs_amd/crs.py::generate_random_points` (tests/test_gpu_whisk.py::test_hash_to_curve_crs_matches_oracle).
#![allow(non_snake_case)]

use ark_bls12_381::{Fr, G1Affine};
use sha2::{Digest, Sha256};

use crate::crs::{CurdleproofsCrs, CRS_EXTRA_POINTS};
use crate::ffi::*;
use crate::N_BLINDERS;

/// |E(Fp)| / r: `mul_by_cofactor` (tests/crs.rs:38) multiplies by the full cofactor, not by h_eff
const COFACTOR: u128 = 0x396c8c005555e1568c00aaab0000aaab;

/// tests/crs.rs:13-52.  increment = 0, 1, ...: hash = SHA-256(seed || be64(increment)); x = `BigInteger384::new` of the hash's four
/// big-endian u64 words as limbs 0..3 (:26-30); keep `get_point_from_x_unchecked(x, false)` (the smaller y, :32-34) times the
/// cofactor if it lies in the subgroup (:35-37).  Candidates go to the device in batches; the verdicts come back per point, so the
/// number of hash attempts equals the reference's.
pub fn generate_random_points(num_points: usize, seed: &str) -> (Vec<G1Affine>, u64) {
    let h = ctx();
    let cofactor = [Fr::from(COFACTOR)]; // < r
    // The candidates are points of E(Fp), not of the order-r subgroup: cpx_g1_scale's default form splits the scalar by the G1
    // endomorphism (an identity of the subgroup only); option scale_any_point selects the plain double-and-add for this context.
    // The context is the thread's shared one (ffi.rs `ctx()`): the option goes back to 0 when this function returns or unwinds, so that
    // later cpx_g1_fold / cpx_g1_scale calls of util_mi355x.rs on this thread keep the 129-step endomorphism form (crs.py: try / finally).
    struct AnyPoint(*mut cpx_ctx, std::ffi::CString);
    impl Drop for AnyPoint {
        fn drop(&mut self) {
            unsafe { cpx_ctx_set_option(self.0, self.1.as_ptr(), 0) };
        }
    }
    let key = std::ffi::CString::new("scale_any_point").unwrap();
    assert_eq!(unsafe { cpx_ctx_set_option(h, key.as_ptr(), 1) }, CPX_OK);
    let _restore = AnyPoint(h, key);
    let mut points: Vec<G1Affine> = vec![];
    let mut increment = 0u64;
    while points.len() != num_points {
        let batch = std::cmp::max(16, 2 * (num_points - points.len()) + 8); // about half of the candidates have a square root
        let mut cands = vec![0u8; 48 * batch];
        for j in 0..batch {
            let mut digest = Sha256::new();
            digest.update(seed.as_bytes());
            digest.update(&(increment + j as u64).to_be_bytes());
            let hash = digest.finalize();
            // x = sum_i word_i * 2^(64 i), word_i = the i-th big-endian u64 of the hash; the compressed encoding is x big-endian in
            // 48 bytes with the compression flag set and the sort flag clear (= the smaller y): limbs 4, 5 are zero
            let c = &mut cands[48 * j..48 * j + 48];
            for i in 0..4 {
                c[48 - 8 * (i + 1)..48 - 8 * i].copy_from_slice(&hash[8 * i..8 * (i + 1)]); // word i, already big-endian
            }
            c[0] |= 0x80;
        }
        let mut aff = vec![0u8; AFF * batch];
        let mut status = vec![0u8; batch];
        let rc = unsafe { cpx_g1_decompress_status(h, cands.as_ptr(), batch, aff.as_mut_ptr(), 0, status.as_mut_ptr()) };
        assert_eq!(rc, CPX_OK, "cpx_g1_decompress_status");
        let hits: Vec<usize> = (0..batch).filter(|j| status[*j] == 0).collect();
        if !hits.is_empty() {
            let mut on_curve = Vec::with_capacity(AFF * hits.len());
            for j in &hits {
                on_curve.extend_from_slice(&aff[AFF * j..AFF * (j + 1)]);
            }
            let mut scaled = vec![0u8; AFF * hits.len()];
            let rc = unsafe { cpx_g1_scale(h, on_curve.as_ptr(), scalars_ptr(&cofactor), 0, hits.len(), scaled.as_mut_ptr()) };
            assert_eq!(rc, CPX_OK, "cpx_g1_scale");
            // is_in_correct_subgroup_assuming_on_curve (:36): compress (cpx_g1_normalize), then the device's subgroup test
            let z = affine_from_wire(&scaled);
            let jac = projective_to_wire(&z.iter().map(|p| (*p).into()).collect::<Vec<_>>());
            let mut comp = vec![0u8; 48 * hits.len()];
            let rc = unsafe { cpx_g1_normalize(h, jac.as_ptr(), hits.len(), std::ptr::null_mut(), comp.as_mut_ptr()) };
            assert_eq!(rc, CPX_OK, "cpx_g1_normalize");
            let mut again = vec![0u8; AFF * hits.len()];
            let mut in_group = vec![0u8; hits.len()];
            let rc = unsafe { cpx_g1_decompress_status(h, comp.as_ptr(), hits.len(), again.as_mut_ptr(), 1, in_group.as_mut_ptr()) };
            assert_eq!(rc, CPX_OK, "cpx_g1_decompress_status");
            for (t, j) in hits.iter().enumerate() {
                if in_group[t] == 0 {
                    points.push(z[t]);
                    if points.len() == num_points {
                        return (points, increment + *j as u64 + 1);
                    }
                }
            }
        }
        increment += batch as u64;
    }
    (points, increment)
}

/// tests/crs.rs:54-66 `ethereum_crs_128_seed`: the CRS of `n` = ell + N_BLINDERS generators from a seed string
pub fn crs_from_seed(ell: usize, seed: &str) -> (CurdleproofsCrs, u64) {
    let (points, attempts) = generate_random_points(ell + N_BLINDERS + CRS_EXTRA_POINTS, seed);
    (CurdleproofsCrs::from_points(ell, &points).unwrap(), attempts)
}
