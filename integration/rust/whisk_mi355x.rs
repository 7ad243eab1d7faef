//! The four byte-level `pub fn`s of `src/whisk.rs` over `cpx_whisk_*` (reference whisk.rs:106-130, :144-179, :183-226,
//! :228-263) under `--features mi355x`: same signatures, same rng draw order, hence the same bytes.  NOT COMPILED here.
#![allow(non_snake_case)]

use ark_bls12_381::Fr;
use ark_serialize::SerializationError;
use ark_std::rand::seq::SliceRandom;
use ark_std::rand::RngCore;
use ark_std::UniformRand;

use crate::crs::CurdleproofsCrs;
use crate::ffi::*;
use crate::util::generate_blinders;
use crate::whisk::{TrackerProofBytes, WhiskShuffleProofBytes, WhiskTracker, ELL, N, TRACKER_PROOF_SIZE, WHISK_SHUFFLE_PROOF_SIZE};
use crate::N_BLINDERS;

fn trackers_to_wire(trackers: &[WhiskTracker]) -> Vec<u8> {
    trackers.iter().flat_map(|t| t.r_G.iter().chain(t.k_r_G.iter()).copied()).collect() // 96 B: r_G || k_r_G (whisk.rs:36-42)
}
fn map_rc(rc: std::os::raw::c_int) -> Result<(), SerializationError> {
    match rc {
        CPX_OK => Ok(()),
        CPX_ERR_DESERIALIZE => Err(SerializationError::InvalidData),
        rc => panic!("libcpx: {}", rc),
    }
}
fn nonzero_factors<T: RngCore>(count: usize, rng: &mut T) -> Vec<Fr> {
    let mut v = Vec::with_capacity(count);
    while v.len() < count {
        let a = Fr::rand(rng);
        if a != Fr::from(0u64) {
            v.push(a);
        }
    }
    v
}

/// whisk.rs:106-130
pub fn is_valid_whisk_shuffle_proof<T: RngCore>(
    rng: &mut T,
    crs: &CurdleproofsCrs,
    pre_trackers: &[WhiskTracker],
    post_trackers: &[WhiskTracker],
    whisk_shuffle_proof_bytes: &WhiskShuffleProofBytes,
) -> Result<bool, SerializationError> {
    let factors = nonzero_factors(8, rng); // the accumulate_check draws of CurdleproofsProof::verify
    let mut valid: std::os::raw::c_int = 0;
    map_rc(unsafe {
        cpx_whisk_is_valid_shuffle_proof(ctx_with_crs(crs), trackers_to_wire(pre_trackers).as_ptr(), trackers_to_wire(post_trackers).as_ptr(),
                                         whisk_shuffle_proof_bytes.as_ptr(), scalars_ptr(&factors), &mut valid)
    })?;
    Ok(valid == 1)
}

/// whisk.rs:144-179
pub fn generate_whisk_shuffle_proof<T: RngCore>(
    rng: &mut T,
    crs: &CurdleproofsCrs,
    pre_trackers: &[WhiskTracker],
) -> Result<(Vec<WhiskTracker>, WhiskShuffleProofBytes), SerializationError> {
    let mut permutation: Vec<u32> = (0..ELL as u32).collect();
    permutation.shuffle(rng); // :152-155
    let k = Fr::rand(rng); // :156
    let vec_m_blinders = generate_blinders(rng, N_BLINDERS); // util.rs:91, inside shuffle_permute_and_commit_input
    let rand: Vec<Fr> = (0..3 * N + 9).map(|_| Fr::rand(rng)).collect(); // the draws of CurdleproofsProof::new
    let mut post = vec![0u8; 96 * ELL];
    let mut proof = [0u8; WHISK_SHUFFLE_PROOF_SIZE];
    map_rc(unsafe {
        cpx_whisk_generate_shuffle_proof(ctx_with_crs(crs), trackers_to_wire(pre_trackers).as_ptr(), permutation.as_ptr(), scalars_ptr(std::slice::from_ref(&k)),
                                         scalars_ptr(&vec_m_blinders), scalars_ptr(&rand), post.as_mut_ptr(), proof.as_mut_ptr())
    })?;
    let post_trackers = post.chunks(96).map(|c| WhiskTracker { r_G: c[..48].try_into().unwrap(), k_r_G: c[48..].try_into().unwrap() }).collect();
    Ok((post_trackers, proof))
}

/// whisk.rs:183-226
pub fn is_valid_whisk_tracker_proof(tracker: &WhiskTracker, k_commitment: &[u8; 48], tracker_proof: &TrackerProofBytes) -> Result<bool, SerializationError> {
    let mut valid: std::os::raw::c_int = 0;
    let t = trackers_to_wire(std::slice::from_ref(tracker));
    map_rc(unsafe { cpx_whisk_is_valid_tracker_proof(ctx(), t.as_ptr(), k_commitment.as_ptr(), tracker_proof.as_ptr(), &mut valid) })?;
    Ok(valid == 1)
}

/// whisk.rs:228-263
pub fn generate_whisk_tracker_proof<T: RngCore>(rng: &mut T, tracker: &WhiskTracker, k: &Fr) -> Result<TrackerProofBytes, SerializationError> {
    let blinder = Fr::rand(rng); // :238
    let mut out = [0u8; TRACKER_PROOF_SIZE];
    let t = trackers_to_wire(std::slice::from_ref(tracker));
    map_rc(unsafe { cpx_whisk_generate_tracker_proof(ctx(), t.as_ptr(), scalars_ptr(std::slice::from_ref(k)), scalars_ptr(std::slice::from_ref(&blinder)), out.as_mut_ptr()) })?;
    Ok(out)
}
