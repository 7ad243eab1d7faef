//! `src/ffi.rs` — bindings to the MI355X core (`include/cpx.h`, `libcpx.so`).  NOT COMPILED in the build image (no rustc there).
//!
//! arkworks 0.4 keeps `Fr` / `Fq` as `BigInt<N>([u64; N])` in Montgomery form and `G1Projective` as `{x, y, z}` of `Fq`:
//! the limb layout of the C-ABI (Fr 32 B, Fp 48 B).  `Fr` is `repr(transparent)`-like over `[u64; 4]` (a newtype chain of single
//! fields), so scalar slices are passed as the memory they occupy (checked once at start-up by `layout_self_check`).  Points are
//! NOT: `G1Affine {x, y, infinity}` and `G1Projective {x, y, z}` are `repr(Rust)` structs whose field order and padding the
//! compiler may choose, so both are marshalled field by field — affine points to 96 bytes `x || y` (the point at infinity as 96 zero
//! bytes), projective points to 144 bytes `X || Y || Z`.
#![allow(non_snake_case, non_camel_case_types, dead_code)]

use ark_bls12_381::{Fq, Fr, G1Affine, G1Projective};
use ark_ff::Zero;
use std::cell::RefCell;
use std::os::raw::{c_char, c_int, c_longlong, c_void};

#[repr(C)]
pub struct cpx_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct cpx_accum {
    _private: [u8; 0],
}

pub const CPX_OK: c_int = 0;
pub const CPX_ERR_ARG: c_int = -1;
pub const CPX_ERR_NOT_POW2: c_int = -2;
pub const CPX_ERR_HIP: c_int = -3;
pub const CPX_ERR_VERIFY: c_int = -4;
pub const CPX_ERR_DESERIALIZE: c_int = -5;
pub const CPX_ERR_STATE: c_int = -6;
pub const CPX_ERR_INTERNAL: c_int = -7;

extern "C" {
    pub fn cpx_ctx_create(device_id: c_int, out: *mut *mut cpx_ctx) -> c_int;
    pub fn cpx_ctx_destroy(ctx: *mut cpx_ctx);
    pub fn cpx_last_error(ctx: *const cpx_ctx) -> *const c_char;
    pub fn cpx_device_count() -> c_int;
    pub fn cpx_ctx_set_option(ctx: *mut cpx_ctx, key: *const c_char, value: c_longlong) -> c_int;
    pub fn cpx_ctx_get_option(ctx: *const cpx_ctx, key: *const c_char, value: *mut c_longlong) -> c_int;
    pub fn cpx_host_alloc(bytes: usize) -> *mut c_void;
    pub fn cpx_host_free(p: *mut c_void);
    // crs.rs:37-58
    pub fn cpx_ctx_set_crs(ctx: *mut cpx_ctx, ell: usize, points: *const u8, n_points: usize) -> c_int;
    pub fn cpx_crs_sums(ctx: *const cpx_ctx, g_sum: *mut u8, h_sum: *mut u8) -> c_int;
    pub fn cpx_proof_size(ctx: *const cpx_ctx) -> usize;
    pub fn cpx_batch_size(ctx: *const cpx_ctx) -> usize;
    // tier 0: util.rs:19-29 and the loops that bypass it
    pub fn cpx_g1_msm(ctx: *mut cpx_ctx, bases: *const u8, scalars: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn cpx_g1_msm_jac(ctx: *mut cpx_ctx, bases: *const u8, scalars: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn cpx_g1_fold(ctx: *mut cpx_ctx, pl: *mut u8, pr: *const u8, gamma: *const u8, half: usize) -> c_int;
    pub fn cpx_g1_scale(ctx: *mut cpx_ctx, p: *const u8, scalars: *const u8, scalar_stride: usize, n: usize, out: *mut u8) -> c_int;
    pub fn cpx_g1_normalize(ctx: *mut cpx_ctx, jac: *const u8, n: usize, out_affine: *mut u8, out_compressed: *mut u8) -> c_int;
    pub fn cpx_g1_decompress(ctx: *mut cpx_ctx, compressed: *const u8, n: usize, out_affine: *mut u8, check_subgroup: c_int) -> c_int;
    // one verdict per point (0 ok, 1 malformed / not on the curve, 2 outside the subgroup): the square root behind tests/crs.rs:13-52
    pub fn cpx_g1_decompress_status(ctx: *mut cpx_ctx, compressed: *const u8, n: usize, out_affine: *mut u8, check_subgroup: c_int, status: *mut u8) -> c_int;
    // tier 1: msm_accumulator.rs:22-68
    pub fn cpx_accum_new(ctx: *mut cpx_ctx, out: *mut *mut cpx_accum) -> c_int;
    pub fn cpx_accum_free(acc: *mut cpx_accum);
    pub fn cpx_accum_check(acc: *mut cpx_accum, c: *const u8, vec_x: *const u8, vec_v: *const u8, n: usize, random_factor: *const u8) -> c_int;
    pub fn cpx_accum_verify(acc: *mut cpx_accum) -> c_int;
    // tier 2: curdleproofs.rs:59, :197 for batches of instances resident in HBM
    pub fn cpx_batch_load(ctx: *mut cpx_ctx, batch: usize, vec_r: *const u8, vec_s: *const u8, vec_t: *const u8, vec_u: *const u8, m: *const u8) -> c_int;
    pub fn cpx_batch_load_begin(ctx: *mut cpx_ctx, batch: usize, vec_r: *const u8, vec_s: *const u8, vec_t: *const u8, vec_u: *const u8, m: *const u8) -> c_int;
    pub fn cpx_batch_load_end(ctx: *mut cpx_ctx) -> c_int;
    pub fn cpx_batch_prove(ctx: *mut cpx_ctx, permutation: *const u32, k: *const u8, vec_m_blinders: *const u8, rand: *const u8, proofs_out: *mut u8) -> c_int;
    pub fn cpx_batch_verify(ctx: *mut cpx_ctx, proofs: *const u8, rand: *const u8, verdict: *mut c_int) -> c_int;
    pub fn cpx_batch_verify_fused(ctx: *mut cpx_ctx, proofs: *const u8, rand: *const u8, partial_jac: *mut u8, n_invalid: *mut c_int) -> c_int;
    pub fn cpx_g1_sum_jac(ctx: *mut cpx_ctx, points_jac: *const u8, n: usize, out_jac: *mut u8, is_identity: *mut c_int) -> c_int;
    // whisk.rs byte-level API
    pub fn cpx_whisk_generate_shuffle_proof(ctx: *mut cpx_ctx, pre_trackers: *const u8, permutation: *const u32, k: *const u8, vec_m_blinders: *const u8,
                                            rand: *const u8, post_trackers_out: *mut u8, proof_out: *mut u8) -> c_int;
    pub fn cpx_whisk_is_valid_shuffle_proof(ctx: *mut cpx_ctx, pre_trackers: *const u8, post_trackers: *const u8, proof: *const u8, rand: *const u8,
                                            valid: *mut c_int) -> c_int;
    pub fn cpx_whisk_generate_tracker_proof(ctx: *mut cpx_ctx, tracker: *const u8, k: *const u8, blinder: *const u8, proof_out: *mut u8) -> c_int;
    pub fn cpx_whisk_is_valid_tracker_proof(ctx: *mut cpx_ctx, tracker: *const u8, k_commitment: *const u8, proof: *const u8, valid: *mut c_int) -> c_int;
}

pub const AFF: usize = 96;
pub const JAC: usize = 144;
pub const FR: usize = 32;

fn fq_bytes(x: &Fq) -> [u8; 48] {
    let mut o = [0u8; 48];
    for (i, limb) in x.0 .0.iter().enumerate() {
        o[8 * i..8 * i + 8].copy_from_slice(&limb.to_le_bytes());
    }
    o
}
fn fq_from_bytes(b: &[u8]) -> Fq {
    let mut limbs = [0u64; 6];
    for i in 0..6 {
        limbs[i] = u64::from_le_bytes(b[8 * i..8 * i + 8].try_into().unwrap());
    }
    ark_ff::Fp(ark_ff::BigInt(limbs), core::marker::PhantomData) // limbs already are the Montgomery representation
}

/// 96 bytes per point, the point at infinity as zeros
pub fn affine_to_wire(points: &[G1Affine]) -> Vec<u8> {
    let mut v = vec![0u8; AFF * points.len()];
    for (i, q) in points.iter().enumerate() {
        if !q.infinity {
            v[AFF * i..AFF * i + 48].copy_from_slice(&fq_bytes(&q.x));
            v[AFF * i + 48..AFF * i + 96].copy_from_slice(&fq_bytes(&q.y));
        }
    }
    v
}
pub fn affine_from_wire(b: &[u8]) -> Vec<G1Affine> {
    b.chunks(AFF)
        .map(|c| {
            if c.iter().all(|x| *x == 0) {
                G1Affine::identity()
            } else {
                G1Affine::new_unchecked(fq_from_bytes(&c[..48]), fq_from_bytes(&c[48..]))
            }
        })
        .collect()
}
/// `&[Fr]` already is wire memory (4 x u64 little-endian Montgomery limbs per element); `layout_self_check` asserts it
pub fn scalars_ptr(s: &[Fr]) -> *const u8 {
    s.as_ptr() as *const u8
}
/// 144 bytes per point, `X || Y || Z`, written field by field (`Projective<P>` is `repr(Rust)`: no cast of the struct's memory)
pub fn projective_to_wire(points: &[G1Projective]) -> Vec<u8> {
    let mut v = vec![0u8; JAC * points.len()];
    for (i, q) in points.iter().enumerate() {
        v[JAC * i..JAC * i + 48].copy_from_slice(&fq_bytes(&q.x));
        v[JAC * i + 48..JAC * i + 96].copy_from_slice(&fq_bytes(&q.y));
        v[JAC * i + 96..JAC * i + 144].copy_from_slice(&fq_bytes(&q.z));
    }
    v
}
/// Run once before the first FFI call (`ctx()` does): what the scalar pointer casts rest on — an `Fr` occupies exactly its four
/// Montgomery limbs, in order, little-endian.
pub fn layout_self_check() {
    use ark_ff::One;
    assert_eq!(std::mem::size_of::<Fr>(), FR, "Fr is not 4 x u64");
    assert_eq!(std::mem::align_of::<Fr>(), 8);
    let x = Fr::one() + Fr::one() + Fr::one();
    let bytes: [u8; FR] = unsafe { std::mem::transmute_copy(&x) };
    for i in 0..4 {
        assert_eq!(u64::from_le_bytes(bytes[8 * i..8 * i + 8].try_into().unwrap()), x.0 .0[i], "Fr memory is not its Montgomery limbs in order");
    }
}
pub fn projective_from_wire(b: &[u8; JAC]) -> G1Projective {
    let z = fq_from_bytes(&b[96..]);
    if z.is_zero() {
        return G1Projective::zero();
    }
    G1Projective::new_unchecked(fq_from_bytes(&b[..48]), fq_from_bytes(&b[48..96]), z)
}

/// One context per thread (a ctx = one device + one stream, not re-entrant); device from `CPX_DEVICE` (default 0).
pub struct Ctx(pub *mut cpx_ctx);
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { cpx_ctx_destroy(self.0) }
    }
}
thread_local! {
    static CTX: RefCell<Option<Ctx>> = RefCell::new(None);
}
pub fn ctx() -> *mut cpx_ctx {
    CTX.with(|c| {
        let mut c = c.borrow_mut();
        if c.is_none() {
            let dev = std::env::var("CPX_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            let mut h: *mut cpx_ctx = std::ptr::null_mut();
            let rc = unsafe { cpx_ctx_create(dev, &mut h) };
            assert_eq!(rc, CPX_OK, "cpx_ctx_create: no usable MI355X (there is no CPU fallback)");
            layout_self_check();
            *c = Some(Ctx(h));
        }
        c.as_ref().unwrap().0
    })
}
/// Loads `crs` into the calling thread's context if it is not the one loaded last (tables are shared process-wide per device).
pub fn ctx_with_crs(crs: &crate::crs::CurdleproofsCrs) -> *mut cpx_ctx {
    thread_local! { static LOADED: RefCell<Vec<u8>> = RefCell::new(Vec::new()); }
    let mut pts: Vec<G1Affine> = Vec::with_capacity(crs.vec_G.len() + 7);
    pts.extend_from_slice(&crs.vec_G);
    pts.extend_from_slice(&crs.vec_H);
    pts.push(crs.H.into());
    pts.push(crs.G_t.into());
    pts.push(crs.G_u.into());
    let wire = affine_to_wire(&pts);
    let h = ctx();
    LOADED.with(|l| {
        let mut l = l.borrow_mut();
        if *l != wire {
            let rc = unsafe { cpx_ctx_set_crs(h, crs.vec_G.len(), wire.as_ptr(), pts.len()) };
            assert_eq!(rc, CPX_OK, "cpx_ctx_set_crs");
            *l = wire;
        }
    });
    h
}
