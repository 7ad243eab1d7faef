// Device-resident protocol steps of the MI355X Curdleproofs core (gfx950) — product code.  See protocol.h.
//
// One 64-lane wave per proof.  The transcript (merlin / STROBE-128, transcript.rs:28-60) runs in the lane-parallel form of
// wave_strobe.hpp — the state bit-interleaved over the two 32-lane halves of the wave, values the callers see uniform over the wave —
// and the scalar-field vectors of a proof (n = ell + 4 entries) are spread over the 64 lanes: elementwise work is
// lane-strided, products and sums are wave reductions, prefix products (the grand-product vector c, powers of beta) are
// chunked scans through LDS.  Every kernel reads the transcript state of its proof, continues it and writes it back;
// nothing here synchronises with the host.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "mont32.hpp"
#include "modinv30.hpp"
#include "strobe.hpp"
#include "wave_strobe.hpp"
#include "protocol.h"

namespace cpx {

#define LBL(s) s, (sizeof(s) - 1)

namespace {

// ---- wave-level helpers on Fr (8 x u32) ----
__device__ __forceinline__ Fr shfl_xor_fr(const Fr& v, int mask) {
  Fr r;
  CPX_UNROLL for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl_xor((int)v.v[i], mask, 64);
  return r;
}
__device__ __forceinline__ Fr shfl_up_fr(const Fr& v, int delta) {
  Fr r;
  CPX_UNROLL for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl_up((int)v.v[i], delta, 64);
  return r;
}
__device__ Fr wave_sum(Fr v) {   // every lane receives the sum over the 64 lanes
  for (int m = 32; m >= 1; m >>= 1) v = fe_add(v, shfl_xor_fr(v, m));
  return v;
}
// inclusive prefix products over a[0..n) in LDS: a[i] <- a[0] * ... * a[i].  64 lanes, lane l owns a contiguous chunk.
__device__ void scan_mul(Fr* a, int n) {
  const int lane = threadIdx.x, chunk = (n + 63) / 64, b = lane * chunk, e = min(n, b + chunk);
  Fr run = Fr::one();
  for (int i = b; i < e; i++) {
    run = fe_mul(run, a[i]);
    a[i] = run;
  }
  Fr t = run;   // inclusive scan of the chunk totals over the lanes
  for (int d = 1; d < 64; d <<= 1) {
    const Fr u = shfl_up_fr(t, d);
    if (lane >= d) t = fe_mul(t, u);
  }
  Fr pre = shfl_up_fr(t, 1);
  if (lane == 0) pre = Fr::one();
  for (int i = b; i < e; i++) a[i] = fe_mul(a[i], pre);
  __syncthreads();
}
__device__ __forceinline__ Fr fr_from_u32(uint32_t x) {
  Fr c = Fr::zero();
  c.v[0] = x;
  return fe_to_mont(c);
}
__device__ __forceinline__ void store_le32(uint8_t* o, const Fr& mont) {   // canonical little-endian bytes (ark-serialize Fr)
  const Fr c = fe_from_mont(mont);
  CPX_UNROLL for (int j = 0; j < 8; j++) {
    o[4 * j] = (uint8_t)c.v[j];
    o[4 * j + 1] = (uint8_t)(c.v[j] >> 8);
    o[4 * j + 2] = (uint8_t)(c.v[j] >> 16);
    o[4 * j + 3] = (uint8_t)(c.v[j] >> 24);
  }
}

struct ProofView {   // pointers of proof p
  const ProveDev& d;
  int p;
  __device__ ProofView(const ProveDev& d_, int p_) : d(d_), p(p_) {}
  __device__ uint8_t* slot(int s) const { return d.slotcomp + ((size_t)p * d.NS + s) * 48; }
  __device__ const Fr* rnd() const { return d.rnd + (size_t)p * (3 * d.n + 9); }
  __device__ Fr* vec(int v) const { return d.vec + ((size_t)p * V_COUNT + v) * d.n; }
  __device__ Fr* sc() const { return d.sc + (size_t)p * SC_COUNT; }
  __device__ uint64_t* tstate() const { return d.tstate + (size_t)p * 27; }
};

}  // namespace

// ------------------------------------------------------------------ prover
// S1 (before phase 1): a_perm = vec_a permuted, followed by the two vec_a_blinders and two zeros — the scalars of
//    A = msm(G | Hvec, a_perm | blinders)  (curdleproofs.rs:85-93)
__global__ __launch_bounds__(64) void k_ps_aperm(const ProveDev d) {
  const ProofView v(d, blockIdx.x);
  const int ell = d.ell;
  const Fr* veca = d.veca + (size_t)v.p * ell;
  const uint32_t* perm = d.perm + (size_t)v.p * ell;
  const RandIdx ri(d.n);
  Fr* ap = v.vec(V_APERM);
  for (int i = threadIdx.x; i < ell; i += 64) ap[i] = veca[perm[i]];
  if (threadIdx.x < 2) ap[ell + threadIdx.x] = v.rnd()[ri.AB() + threadIdx.x];
  else if (threadIdx.x < 4) ap[ell + threadIdx.x] = Fr::zero();
}

// S2 (after phase 1: A, B_a, B_t, B_u, B_c, the T_1 commitments): SamePerm challenges, the grand-product factors and the
//    partial-product vector c.  same_permutation_argument.rs:60-83, grand_product_argument.rs:66-75
__global__ __launch_bounds__(64) void k_ps_sameperm(const ProveDev d) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  Fr* buf = reinterpret_cast<Fr*>(smem);            // [n]
  uint8_t* scratch = smem + (size_t)d.n * 32;       // [64]
  const ProofView v(d, blockIdx.x);
  const int lane = threadIdx.x, ell = d.ell;
  const RandIdx ri(d.n);
  const Fr* veca = d.veca + (size_t)v.p * ell;
  const uint32_t* perm = d.perm + (size_t)v.p * ell;
  WaveStrobe t;
  t.load(v.tstate(), lane);
  t.append_message(LBL("same_perm_step1"), v.slot(SL_A), 48, scratch);
  t.append_message(LBL("same_perm_step1"), d.mcomp + (size_t)v.p * 48, 48, scratch);
  // vec_a as Vec<Fr>: u64-LE length, then the canonical scalars, one message
  uint32_t* w = reinterpret_cast<uint32_t*>(buf);
  for (int i = lane; i < ell; i += 64) {
    const Fr c = fe_from_mont(veca[i]);
    CPX_UNROLL for (int j = 0; j < 8; j++) w[8 * i + j] = c.v[j];
  }
  if (lane < 8) scratch[8 + lane] = (uint8_t)((uint64_t)ell >> (8 * lane));
  __syncthreads();
  t.append_begin(LBL("same_perm_step1"), 8 + 32 * (size_t)ell, scratch);
  t.absorb(scratch + 8, 8);
  t.absorb(reinterpret_cast<const uint8_t*>(buf), 32 * (size_t)ell);
  const Fr alpha = t.challenge_scalar(LBL("same_perm_alpha"), scratch);
  const Fr beta = t.challenge_scalar(LBL("same_perm_beta"), scratch);
  __syncthreads();
  // factors_i = a_sigma(i) + sigma(i) alpha + beta; gprod = prod factors; c = exclusive prefix products | vec_c_blinders
  const Fr* ap = v.vec(V_APERM);
  Fr* fact = v.vec(V_FACT);
  for (int i = lane; i < ell; i += 64) {
    const Fr f = fe_add(fe_add(ap[i], fe_mul(fr_from_u32(perm[i]), alpha)), beta);
    fact[i] = f;
    buf[i] = f;
  }
  // (entries ell .. ell + 3: the scalars of B = A + alpha M + beta sum(G) on the blinder bases Hvec, a_blinders | 0 0 + alpha vec_m_blinders —
  // with them V_FACT is the scalar vector of B as ONE multi-scalar multiplication over G | Hvec, the form small batches take)
  if (lane < 4) fact[ell + lane] = fe_add(ap[ell + lane], fe_mul(d.mbl[(size_t)v.p * 4 + lane], alpha));
  __syncthreads();
  scan_mul(buf, ell);
  Fr* c = v.vec(V_C);
  for (int i = lane; i < ell; i += 64) c[i] = i ? buf[i - 1] : Fr::one();
  if (lane < 4) c[ell + lane] = v.rnd()[ri.CB() + lane];
  if (lane == 0) {
    Fr* sc = v.sc();
    sc[SC_ALPHA_SP] = alpha;
    sc[SC_BETA_SP] = beta;
    sc[SC_GPROD] = buf[ell - 1];
  }
  t.store(v.tstate());
}

// S3 (after phase 2: B, A', C): grand-product challenges, the IPA witness vectors and blinders.
//    grand_product_argument.rs:63-65,76-132, inner_product_argument.rs:42-82 (generate_ipa_blinders)
__global__ __launch_bounds__(64) void k_ps_gprod(const ProveDev d) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  Fr* buf = reinterpret_cast<Fr*>(smem);
  uint8_t* scratch = smem + (size_t)d.n * 32;
  const ProofView v(d, blockIdx.x);
  const int lane = threadIdx.x, ell = d.ell, n = d.n;
  const RandIdx ri(n);
  const Fr* rnd = v.rnd();
  Fr* sc = v.sc();
  const Fr alpha_sp = sc[SC_ALPHA_SP], gprod = sc[SC_GPROD];
  WaveStrobe t;
  t.load(v.tstate(), lane);
  t.append_message(LBL("gprod_step1"), v.slot(SL_B), 48, scratch);
  t.append_scalar(LBL("gprod_step1"), gprod, scratch);
  const Fr alpha_g = t.challenge_scalar(LBL("gprod_alpha"), scratch);
  // r_b + alpha (the blinders of B = A + alpha_sp M + ..., shifted by alpha_g) and r_p = <r_b + alpha, vec_c_blinders>
  Fr rpa[4], r_p = Fr::zero();
  CPX_UNROLL for (int i = 0; i < 4; i++) {
    const Fr ab = i < 2 ? rnd[ri.AB() + i] : Fr::zero();
    rpa[i] = fe_add(fe_add(ab, fe_mul(alpha_sp, d.mbl[(size_t)v.p * 4 + i])), alpha_g);
    r_p = fe_add(r_p, fe_mul(rpa[i], rnd[ri.CB() + i]));
  }
  t.append_message(LBL("gprod_step2"), v.slot(SL_C), 48, scratch);
  t.append_scalar(LBL("gprod_step2"), r_p, scratch);
  const Fr beta = t.challenge_scalar(LBL("gprod_beta"), scratch);
  t.store(v.tstate());
  const Fr beta_inv = fr_inv_divsteps(beta);
  __syncthreads();
  // powers of beta: buf[i] = beta^(i+1), i <= ell
  for (int i = lane; i <= ell; i += 64) buf[i] = beta;
  __syncthreads();
  scan_mul(buf, ell + 1);
  const Fr* fact = v.vec(V_FACT);
  Fr* dd = v.vec(V_D);
  for (int i = lane; i < ell; i += 64) dd[i] = fe_sub(fe_mul(fact[i], buf[i]), i ? buf[i - 1] : Fr::one());
  const Fr beta_l = buf[ell - 1], beta_l1 = buf[ell];
  if (lane < 4) dd[ell + lane] = fe_mul(beta_l1, rpa[lane]);
  const Fr z_ip = fe_sub(fe_add(fe_mul(r_p, beta_l1), fe_mul(gprod, beta_l)), Fr::one());
  __syncthreads();
  // u_i = beta^-(i+1) for i < ell, beta^-(ell+1) on the blinder positions (grand_product_argument.rs:90-102)
  for (int i = lane; i <= ell; i += 64) buf[i] = beta_inv;
  __syncthreads();
  scan_mul(buf, ell + 1);
  Fr* u = v.vec(V_U);
  for (int i = lane; i < n; i += 64) u[i] = buf[i < ell ? i : ell];
  __syncthreads();
  // generate_ipa_blinders: r = rnd[IR..], z = rnd[IZ..] with the last two entries solved for
  const Fr* c = v.vec(V_C);
  const Fr* r = rnd + ri.IR();
  const Fr* z = rnd + ri.IZ();
  Fr om = Fr::zero(), de = Fr::zero();
  for (int i = lane; i < n; i += 64) {
    om = fe_add(om, fe_mul(r[i], dd[i]));
    if (i < n - 2) {
      om = fe_add(om, fe_mul(z[i], c[i]));
      de = fe_add(de, fe_mul(r[i], z[i]));
    }
  }
  const Fr omega = wave_sum(om), delta = wave_sum(de);
  const Fr inv_c = fr_inv_divsteps(c[n - 2]);
  const Fr den = fe_add(fe_mul(fe_mul(fe_neg(r[n - 2]), inv_c), c[n - 1]), r[n - 1]);
  const Fr last_z = fe_mul(fe_sub(fe_mul(fe_mul(r[n - 2], inv_c), omega), delta), fr_inv_divsteps(den));
  const Fr pen_z = fe_mul(fe_neg(inv_c), fe_add(fe_mul(last_z, c[n - 1]), omega));
  Fr* zz = v.vec(V_ZZ);
  Fr* zzu = v.vec(V_ZZU);
  for (int i = lane; i < n; i += 64) {
    const Fr zi = i < n - 2 ? z[i] : (i == n - 2 ? pen_z : last_z);
    zz[i] = zi;
    zzu[i] = fe_mul(zi, u[i]);   // B_d = msm(G', r_d) = msm(G, r_d o u)
  }
  if (lane == 0) {
    sc[SC_ALPHA_G] = alpha_g;
    sc[SC_BETA_G] = beta;
    sc[SC_BETA_G_INV] = beta_inv;
    sc[SC_NEG_BETA_G_INV] = fe_neg(beta_inv);
    sc[SC_RP] = r_p;
    sc[SC_ZIP] = z_ip;
  }
}

// S4 (after phase 3: D, B_d): IPA challenges alpha, beta; the round vectors c | d | S_G | S_G' move to their device
//    home.  inner_product_argument.rs:129-148
__global__ __launch_bounds__(64) void k_ps_ipa_setup(const ProveDev d) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* scratch = smem;
  const ProofView v(d, blockIdx.x);
  const int lane = threadIdx.x, n = d.n;
  const SlotMap sm(d.L);
  const RandIdx ri(n);
  Fr* sc = v.sc();
  WaveStrobe t;
  t.load(v.tstate(), lane);
  t.append_message(LBL("ipa_step1"), v.slot(SL_C), 48, scratch);
  t.append_message(LBL("ipa_step1"), v.slot(sm.D()), 48, scratch);
  t.append_scalar(LBL("ipa_step1"), sc[SC_ZIP], scratch);
  t.append_message(LBL("ipa_step1"), v.slot(SL_BC), 48, scratch);
  t.append_message(LBL("ipa_step1"), v.slot(SL_BD), 48, scratch);
  const Fr alpha = t.challenge_scalar(LBL("ipa_alpha"), scratch);
  const Fr beta = t.challenge_scalar(LBL("ipa_beta"), scratch);
  t.store(v.tstate());
  const Fr* r = v.rnd() + ri.IR();
  const Fr *c = v.vec(V_C), *dd = v.vec(V_D), *zz = v.vec(V_ZZ), *u = v.vec(V_U);
  Fr* rv = d.rvec + (size_t)v.p * 4 * n;
  for (int i = lane; i < n; i += 64) {
    rv[i] = fe_add(r[i], fe_mul(alpha, c[i]));
    rv[n + i] = fe_add(zz[i], fe_mul(alpha, dd[i]));
    rv[2 * n + i] = Fr::one();
    rv[3 * n + i] = u[i];
  }
  if (lane == 0) {
    sc[SC_ALPHA_I] = alpha;
    sc[SC_BETA_I] = beta;
    d.rbeta[v.p] = beta;
  }
}

// S5.j (after round j's four cross terms): hash L_C, L_D, R_C, R_D, draw gamma; gamma and gamma^-1 go to the fold kernel.
//    inner_product_argument.rs:164-171
__global__ __launch_bounds__(64) void k_ps_ipa_round(const ProveDev d, int j) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* scratch = smem;
  const ProofView v(d, blockIdx.x);
  const int lane = threadIdx.x;
  const SlotMap sm(d.L);
  WaveStrobe t;
  t.load(v.tstate(), lane);
  t.append_message(LBL("ipa_loop"), v.slot(sm.LC(j)), 48, scratch);
  t.append_message(LBL("ipa_loop"), v.slot(sm.LD(j)), 48, scratch);
  t.append_message(LBL("ipa_loop"), v.slot(sm.RC(j)), 48, scratch);
  t.append_message(LBL("ipa_loop"), v.slot(sm.RD(j)), 48, scratch);
  const Fr gamma = t.challenge_scalar(LBL("ipa_gamma"), scratch);
  t.store(v.tstate());
  const Fr gi = fr_inv_divsteps(gamma);
  if (lane == 0) {
    d.rgam[2 * (size_t)v.p] = gamma;
    d.rgam[2 * (size_t)v.p + 1] = gi;
  }
}

// S6 (after the last IPA fold; R, S and the T_2 commitments have arrived from the side stream): SameScalar responses,
//    SameMSM step 1 and its witness vector x.  same_scalar_argument.rs:63-77, curdleproofs.rs:134-169,
//    same_multiscalar_argument.rs:84-97
__global__ __launch_bounds__(64) void k_ps_smsm_setup(const ProveDev d) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* scratch = smem;         // [64]
  uint8_t* tail = smem + 64;       // [4 * 48] blinder slots of vec_T / vec_U
  const ProofView v(d, blockIdx.x);
  const int lane = threadIdx.x, ell = d.ell, n = d.n;
  const SlotMap sm(d.L);
  const RandIdx ri(n);
  const Fr* rnd = v.rnd();
  Fr* sc = v.sc();
  if (lane == 0) {   // c[0], d[0] after the last fold (inner_product_argument.rs:188-195)
    sc[SC_CFIN] = d.rvec[(size_t)v.p * 4 * n];
    sc[SC_DFIN] = d.rvec[(size_t)v.p * 4 * n + n];
  }
  WaveStrobe t;
  t.load(v.tstate(), lane);
  const int pts[10] = {SL_R, SL_S, SL_CMT1, SL_CMT2, SL_CMU1, SL_CMU2, sm.CMA1(), sm.CMA2(), sm.CMB1(), sm.CMB2()};
  for (int q = 0; q < 10; q++) t.append_message(LBL("sameexp_points"), v.slot(pts[q]), 48, scratch);
  const Fr alpha_s = t.challenge_scalar(LBL("same_scalar_alpha"), scratch);
  const Fr z_k = fe_add(rnd[ri.RK()], fe_mul(d.k[v.p], alpha_s));
  const Fr z_t = fe_add(rnd[ri.RA()], fe_mul(rnd[ri.RT()], alpha_s));
  const Fr z_u = fe_add(rnd[ri.RB()], fe_mul(rnd[ri.RU()], alpha_s));
  t.append_message(LBL("same_msm_step1"), v.slot(sm.APRIME()), 48, scratch);
  t.append_message(LBL("same_msm_step1"), v.slot(SL_CMT2), 48, scratch);
  t.append_message(LBL("same_msm_step1"), v.slot(SL_CMU2), 48, scratch);
  // vec_T || O O H O and vec_U || O O O H as Vec<G1Affine> (curdleproofs.rs:141-155)
  const uint8_t* ic = d.inst_comp + (size_t)v.p * 4 * ell * 48;
  for (int which = 0; which < 2; which++) {
    for (int i = lane; i < 4 * 48; i += 64) {
      const int s = i / 48, b = i % 48;
      const bool h = s == (which ? 3 : 2);
      tail[i] = h ? d.crs_h_comp[b] : (b == 0 ? 0xc0 : 0);
    }
    if (lane < 8) scratch[8 + lane] = (uint8_t)((uint64_t)n >> (8 * lane));
    __syncthreads();
    t.append_begin(LBL("same_msm_step1"), 8 + 48 * (size_t)n, scratch);
    t.absorb(scratch + 8, 8);
    t.absorb(ic + (size_t)(2 + which) * ell * 48, 48 * (size_t)ell);
    t.absorb(tail, 4 * 48);
    __syncthreads();
  }
  t.append_message(LBL("same_msm_step1"), v.slot(sm.BA()), 48, scratch);
  t.append_message(LBL("same_msm_step1"), v.slot(sm.BT()), 48, scratch);
  t.append_message(LBL("same_msm_step1"), v.slot(sm.BU()), 48, scratch);
  const Fr alpha_m = t.challenge_scalar(LBL("same_msm_alpha"), scratch);
  t.store(v.tstate());
  // x = vec_r + alpha (a_perm | a_blinders | r_t | r_u); fold coefficients S_M = 1
  const Fr* vr = rnd + ri.VR();
  const Fr* ap = v.vec(V_APERM);
  Fr* rv = d.rvec2 + (size_t)v.p * 2 * n;
  for (int i = lane; i < n; i += 64) {
    const Fr w = i < ell + 2 ? ap[i] : (i == ell + 2 ? rnd[ri.RT()] : rnd[ri.RU()]);
    rv[i] = fe_add(vr[i], fe_mul(alpha_m, w));
    rv[n + i] = Fr::one();
  }
  if (lane == 0) {
    sc[SC_ALPHA_S] = alpha_s;
    sc[SC_ALPHA_M] = alpha_m;
    sc[SC_ZK] = z_k;
    sc[SC_ZT] = z_t;
    sc[SC_ZU] = z_u;
  }
}

// S7.j: hash L_A, L_T, L_U, R_A, R_T, R_U of SameMSM round j, draw gamma.  same_multiscalar_argument.rs:114-123
__global__ __launch_bounds__(64) void k_ps_smsm_round(const ProveDev d, int j) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* scratch = smem;
  const ProofView v(d, blockIdx.x);
  const int lane = threadIdx.x;
  const SlotMap sm(d.L);
  WaveStrobe t;
  t.load(v.tstate(), lane);
  const int six[6] = {sm.LA(j), sm.LT(j), sm.LU(j), sm.RA(j), sm.RT(j), sm.RU(j)};
  for (int q = 0; q < 6; q++) t.append_message(LBL("same_msm_loop"), v.slot(six[q]), 48, scratch);
  const Fr gamma = t.challenge_scalar(LBL("same_msm_gamma"), scratch);
  t.store(v.tstate());
  const Fr gi = fr_inv_divsteps(gamma);
  if (lane == 0) {
    d.rgam[2 * (size_t)v.p] = gamma;
    d.rgam[2 * (size_t)v.p + 1] = gi;
  }
}

// S8: CurdleproofsProof::serialize (curdleproofs.rs:300-310 and the sub-proofs' serialisers): the proof points in slot
//     order with the seven scalars interleaved
__global__ __launch_bounds__(64) void k_ps_serialize(const ProveDev d) {
  const ProofView v(d, blockIdx.x);
  const int lane = threadIdx.x, L = d.L, n = d.n;
  const SlotMap sm(L);
  uint8_t* o = d.proofs + (size_t)v.p * d.psz;
  const int npp = sm.n_proof_points();
  const int b1 = 9, b2 = 11 + 4 * L, b3 = 15 + 4 * L;   // proof points before r_p / before c, d / before z_k, z_t, z_u
  for (int i = lane; i < npp * 12; i += 64) {           // 48 bytes = 12 words per point
    const int q = i / 12, wd = i % 12;
    const size_t off = 48 * (size_t)q + (q >= b1 ? 32 : 0) + (q >= b2 ? 64 : 0) + (q >= b3 ? 96 : 0);
    const uint8_t* src = v.slot(SL_A + q) + 4 * wd;
    uint8_t* dst = o + off + 4 * wd;
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
    dst[3] = src[3];
  }
  const Fr* sc = v.sc();
  if (lane < 7) {
    const size_t off_rp = 48 * (size_t)b1, off_c = off_rp + 32 + 48 * (size_t)(b2 - b1), off_zk = off_c + 64 + 48 * (size_t)(b3 - b2),
                 off_x = off_zk + 96 + 48 * (size_t)(npp - b3);
    const size_t offs[7] = {off_rp, off_c, off_c + 32, off_zk, off_zk + 32, off_zk + 64, off_x};
    const Fr xf = d.rvec2[(size_t)v.p * 2 * n];   // x[0] after the last fold (same_multiscalar_argument.rs:138-141)
    const Fr vals[7] = {sc[SC_RP], sc[SC_CFIN], sc[SC_DFIN], sc[SC_ZK], sc[SC_ZT], sc[SC_ZU], xf};
    store_le32(o + offs[lane], vals[lane]);
  }
}

// ------------------------------------------------------------------ verifier
namespace {
struct VerifyView {
  const VerifyDev& d;
  int p;
  __device__ VerifyView(const VerifyDev& d_, int p_) : d(d_), p(p_) {}
  __device__ const uint8_t* pb() const { return d.proofs + (size_t)p * d.psz; }
  // byte offset of proof point q (slot SL_A + q) inside the serialized proof: scalars are interleaved after 9, 11 + 4L and 15 + 4L points
  __device__ size_t pt_off(int q) const { return 48 * (size_t)q + (q >= 9 ? 32 : 0) + (q >= 11 + 4 * d.L ? 64 : 0) + (q >= 15 + 4 * d.L ? 96 : 0); }
  __device__ const uint8_t* P(int slot) const { return pb() + pt_off(slot - SL_A); }
  __device__ size_t off_rp() const { return 48 * 9; }
  __device__ size_t off_c() const { return off_rp() + 32 + 48 * (size_t)(2 + 4 * d.L); }
  __device__ size_t off_zk() const { return off_c() + 64 + 48 * 4; }
  __device__ size_t off_x() const { return off_zk() + 96 + 48 * (size_t)(3 + 6 * d.L); }
  __device__ uint8_t* slot(int s) const { return d.slotcomp + ((size_t)p * d.NS + s) * 48; }
  __device__ Fr* vsc() const { return d.vsc + (size_t)p * VSC_COUNT; }
  __device__ uint64_t* tstate() const { return d.tstate + (size_t)p * 27; }
  __device__ const Fr* rnd() const { return d.rnd + (size_t)p * d.rand_stride; }
};
// canonical little-endian scalar -> Montgomery; false (and zero) if >= r  (Fr::deserialize_compressed)
__device__ bool load_scalar(const uint8_t* b, Fr& out) {
  Fr c;
  CPX_UNROLL for (int j = 0; j < 8; j++) c.v[j] = (uint32_t)b[4 * j] | ((uint32_t)b[4 * j + 1] << 8) | ((uint32_t)b[4 * j + 2] << 16) | ((uint32_t)b[4 * j + 3] << 24);
  bool lt = false;
  for (int j = 7; j >= 0; j--) {
    if (c.v[j] != FrCfg::P[j]) {
      lt = c.v[j] < FrCfg::P[j];
      break;
    }
  }
  out = lt ? fe_to_mont(c) : Fr::zero();
  return lt;
}
__device__ Fr fr_pow_u32(const Fr& a, uint32_t e) {
  Fr r = Fr::one();
  for (int i = 31; i >= 0; i--) {
    r = fe_mul(r, r);
    if ((e >> i) & 1) r = fe_mul(r, a);
  }
  return r;
}
}  // namespace

// V1 (after decompression of the proof points and the transcript prefix): deserialisation verdict, the transcript up to the
//    grand-product beta, the scalars of D.  curdleproofs.rs:213-222, same_permutation_argument.rs:131-145,
//    grand_product_argument.rs:200-223
__global__ __launch_bounds__(64) void k_vs_prefix(const VerifyDev d) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  Fr* buf = reinterpret_cast<Fr*>(smem);
  uint8_t* scratch = smem + (size_t)d.n * 32;
  const VerifyView v(d, blockIdx.x);
  const int lane = threadIdx.x, ell = d.ell;
  const SlotMap sm(d.L);
  const uint8_t* pb = v.pb();
  // the seven scalars of the proof must be canonical, every proof point must have decoded (status != 0: malformed / off the curve / outside the subgroup)
  Fr r_p, c_fin, d_fin, z_k, z_t, z_u, x_fin;
  bool ok = load_scalar(pb + v.off_rp(), r_p);
  ok &= load_scalar(pb + v.off_c(), c_fin);
  ok &= load_scalar(pb + v.off_c() + 32, d_fin);
  ok &= load_scalar(pb + v.off_zk(), z_k);
  ok &= load_scalar(pb + v.off_zk() + 32, z_t);
  ok &= load_scalar(pb + v.off_zk() + 64, z_u);
  ok &= load_scalar(pb + v.off_x(), x_fin);
  const int npp = sm.n_proof_points();
  int badpt = 0;
  for (int q = lane; q < npp; q += 64) badpt |= d.status[(size_t)v.p * npp + q];
  const bool bad = !ok || __any(badpt != 0);
  const uint8_t* ic = d.inst_comp + (size_t)v.p * 4 * ell * 48;
  const bool reject = ic[(size_t)2 * ell * 48] == 0xc0;   // curdleproofs.rs:218: the randomiser must not have wiped the ciphertexts
  const Fr* veca = d.veca + (size_t)v.p * ell;
  WaveStrobe t;
  t.load(v.tstate(), lane);
  t.append_message(LBL("same_perm_step1"), v.P(SL_A), 48, scratch);
  t.append_message(LBL("same_perm_step1"), d.mcomp + (size_t)v.p * 48, 48, scratch);
  uint32_t* w = reinterpret_cast<uint32_t*>(buf);
  for (int i = lane; i < ell; i += 64) {
    const Fr c = fe_from_mont(veca[i]);
    CPX_UNROLL for (int j = 0; j < 8; j++) w[8 * i + j] = c.v[j];
  }
  if (lane < 8) scratch[8 + lane] = (uint8_t)((uint64_t)ell >> (8 * lane));
  __syncthreads();
  t.append_begin(LBL("same_perm_step1"), 8 + 32 * (size_t)ell, scratch);
  t.absorb(scratch + 8, 8);
  t.absorb(reinterpret_cast<const uint8_t*>(buf), 32 * (size_t)ell);
  const Fr alpha_sp = t.challenge_scalar(LBL("same_perm_alpha"), scratch);
  const Fr beta_sp = t.challenge_scalar(LBL("same_perm_beta"), scratch);
  __syncthreads();
  // gprod = prod (a_i + i alpha + beta)
  for (int i = lane; i < ell; i += 64) buf[i] = fe_add(fe_add(veca[i], fe_mul(fr_from_u32((uint32_t)i), alpha_sp)), beta_sp);
  __syncthreads();
  scan_mul(buf, ell);
  const Fr gprod = buf[ell - 1];
  t.append_message(LBL("gprod_step1"), v.P(SL_B), 48, scratch);
  t.append_scalar(LBL("gprod_step1"), gprod, scratch);
  const Fr alpha_g = t.challenge_scalar(LBL("gprod_alpha"), scratch);
  t.append_message(LBL("gprod_step2"), v.P(SL_C), 48, scratch);
  t.append_scalar(LBL("gprod_step2"), r_p, scratch);
  const Fr beta_g = t.challenge_scalar(LBL("gprod_beta"), scratch);
  t.store(v.tstate());
  const Fr beta_g_inv = fr_inv_divsteps(beta_g);
  if (lane == 0) {
    Fr* sc = v.vsc();
    sc[VSC_NEG_BETA_G_INV] = fe_neg(beta_g_inv);
    sc[VSC_ALPHA_G] = alpha_g;
    sc[VSC_ALPHA_SP] = alpha_sp;
    sc[VSC_BETA_SP] = beta_sp;
    sc[VSC_GPROD] = gprod;
    sc[VSC_BETA_G] = beta_g;
    sc[VSC_BETA_G_INV] = beta_g_inv;
    sc[VSC_RP] = r_p;
    sc[VSC_CFIN] = c_fin;
    sc[VSC_DFIN] = d_fin;
    sc[VSC_ZK] = z_k;
    sc[VSC_ZT] = z_t;
    sc[VSC_ZU] = z_u;
    sc[VSC_XFIN] = x_fin;
    d.flags[v.p] = (bad ? 1u : 0u) | (reject ? 2u : 0u);
  }
}

// V2 (after D and A' have been computed and compressed): the rest of the transcript and the scalars of the flattened
//    accumulated check  sum_i a_i (lhs_i - x_i . V_i) == O  (msm_accumulator.rs:38-68):
//    inner_product_argument.rs:202-326, same_scalar_argument.rs:112-137, same_multiscalar_argument.rs:153-261,
//    grand_product_argument.rs:211-246, same_permutation_argument.rs:146-171, curdleproofs.rs:283-297.
//    scal[p] = [ R | S | T | U (4 ell) | misc: CRS singles, M, every proof point (NM) ],  scal_crs[p] = [ G | Hvec (n) ]
__global__ __launch_bounds__(64) void k_vs_scalars(const VerifyDev d) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  Fr* buf = reinterpret_cast<Fr*>(smem);                       // [n]
  Fr* gam = buf + d.n;                                         // [4 L]: gamma_i | gamma_m | their inverses
  uint8_t* scratch = reinterpret_cast<uint8_t*>(gam + 4 * d.L);   // [64]
  uint8_t* tail = scratch + 64;                                // [4 * 48]
  const VerifyView v(d, blockIdx.x);
  const int lane = threadIdx.x, ell = d.ell, n = d.n, L = d.L;
  const SlotMap sm(L);
  const Fr* sc = v.vsc();
  const Fr alpha_sp = sc[VSC_ALPHA_SP], beta_sp = sc[VSC_BETA_SP], gprod = sc[VSC_GPROD], alpha_g = sc[VSC_ALPHA_G], beta_g = sc[VSC_BETA_G],
           beta_g_inv = sc[VSC_BETA_G_INV], r_p = sc[VSC_RP], c_fin = sc[VSC_CFIN], d_fin = sc[VSC_DFIN], z_k = sc[VSC_ZK], z_t = sc[VSC_ZT],
           z_u = sc[VSC_ZU], x_fin = sc[VSC_XFIN];
  const Fr beta_l = fr_pow_u32(beta_g, (uint32_t)ell), beta_l1 = fe_mul(beta_l, beta_g);
  const Fr z_ip = fe_sub(fe_add(fe_mul(r_p, beta_l1), fe_mul(gprod, beta_l)), Fr::one());
  WaveStrobe t;
  t.load(v.tstate(), lane);
  t.append_message(LBL("ipa_step1"), v.P(SL_C), 48, scratch);
  t.append_message(LBL("ipa_step1"), v.slot(sm.D()), 48, scratch);
  t.append_scalar(LBL("ipa_step1"), z_ip, scratch);
  t.append_message(LBL("ipa_step1"), v.P(SL_BC), 48, scratch);
  t.append_message(LBL("ipa_step1"), v.P(SL_BD), 48, scratch);
  const Fr alpha_i = t.challenge_scalar(LBL("ipa_alpha"), scratch);
  const Fr beta_i = t.challenge_scalar(LBL("ipa_beta"), scratch);
  for (int j = 0; j < L; j++) {
    t.append_message(LBL("ipa_loop"), v.P(sm.LC(j)), 48, scratch);
    t.append_message(LBL("ipa_loop"), v.P(sm.LD(j)), 48, scratch);
    t.append_message(LBL("ipa_loop"), v.P(sm.RC(j)), 48, scratch);
    t.append_message(LBL("ipa_loop"), v.P(sm.RD(j)), 48, scratch);
    const Fr g = t.challenge_scalar(LBL("ipa_gamma"), scratch);
    if (lane == 0) gam[j] = g;
  }
  {
    const int sp[10] = {SL_R, SL_S, SL_CMT1, SL_CMT2, SL_CMU1, SL_CMU2, sm.CMA1(), sm.CMA2(), sm.CMB1(), sm.CMB2()};
    for (int q = 0; q < 10; q++) t.append_message(LBL("sameexp_points"), v.P(sp[q]), 48, scratch);
  }
  const Fr alpha_s = t.challenge_scalar(LBL("same_scalar_alpha"), scratch);
  t.append_message(LBL("same_msm_step1"), v.slot(sm.APRIME()), 48, scratch);
  t.append_message(LBL("same_msm_step1"), v.P(SL_CMT2), 48, scratch);
  t.append_message(LBL("same_msm_step1"), v.P(SL_CMU2), 48, scratch);
  const uint8_t* ic = d.inst_comp + (size_t)v.p * 4 * ell * 48;
  for (int which = 0; which < 2; which++) {   // vec_T || O O H O and vec_U || O O O H (curdleproofs.rs:262-276)
    for (int i = lane; i < 4 * 48; i += 64) {
      const int s = i / 48, b = i % 48;
      const bool h = s == (which ? 3 : 2);
      tail[i] = h ? d.crs_h_comp[b] : (b == 0 ? 0xc0 : 0);
    }
    if (lane < 8) scratch[8 + lane] = (uint8_t)((uint64_t)n >> (8 * lane));
    __syncthreads();
    t.append_begin(LBL("same_msm_step1"), 8 + 48 * (size_t)n, scratch);
    t.absorb(scratch + 8, 8);
    t.absorb(ic + (size_t)(2 + which) * ell * 48, 48 * (size_t)ell);
    t.absorb(tail, 4 * 48);
    __syncthreads();
  }
  t.append_message(LBL("same_msm_step1"), v.P(sm.BA()), 48, scratch);
  t.append_message(LBL("same_msm_step1"), v.P(sm.BT()), 48, scratch);
  t.append_message(LBL("same_msm_step1"), v.P(sm.BU()), 48, scratch);
  const Fr alpha_m = t.challenge_scalar(LBL("same_msm_alpha"), scratch);
  for (int j = 0; j < L; j++) {
    const int six[6] = {sm.LA(j), sm.LT(j), sm.LU(j), sm.RA(j), sm.RT(j), sm.RU(j)};
    for (int q = 0; q < 6; q++) t.append_message(LBL("same_msm_loop"), v.P(six[q]), 48, scratch);
    const Fr g = t.challenge_scalar(LBL("same_msm_gamma"), scratch);
    if (lane == 0) gam[L + j] = g;
  }
  t.store(v.tstate());
  __syncthreads();
  // the 2 L challenge inverses, one lane each (ark_ff::batch_inversion of the reference gives the same values)
  if (lane < 2 * L) gam[2 * L + lane] = fr_inv_divsteps(gam[lane]);
  __syncthreads();
  const Fr *gam_i = gam, *gam_m = gam + L, *gam_i_inv = gam + 2 * L, *gam_m_inv = gam + 3 * L;
  // u_i = beta^-(i+1), i < ell; beta^-(ell+1) on the blinder positions (grand_product_argument.rs:211-219)
  for (int i = lane; i <= ell; i += 64) buf[i] = beta_g_inv;
  __syncthreads();
  scan_mul(buf, ell + 1);
  const Fr* rnd = v.rnd();
  const Fr a1 = rnd[0], a2 = rnd[1], a3 = rnd[2], a4 = rnd[3], a5 = rnd[4], a6 = rnd[5], a7 = rnd[6], a8 = rnd[7];
  const Fr a2c = fe_mul(a2, c_fin), a3d = fe_mul(a3, d_fin), a4x = fe_mul(a4, x_fin), a5x = fe_mul(a5, x_fin), a6x = fe_mul(a6, x_fin);
  const Fr a1b = fe_mul(a1, beta_sp);
  const bool bad = d.flags[v.p] & 1u;   // an undecodable proof contributes nothing to a fused batch (it is counted as invalid)
  const size_t NI = 4 * (size_t)ell, NPT = NI + d.NM;
  Fr* out = d.scal + (size_t)v.p * NPT;
  Fr* out_crs = d.scal_crs + (size_t)v.p * n;
  // verification scalars s_i = prod_{j : bit (L-1-j) of i set} gamma_j  (util.rs:40-64)
  auto svec = [&](const Fr* g, int i) {
    Fr r = Fr::one();
    for (int j = 0; j < L; j++)
      if ((i >> (L - 1 - j)) & 1) r = fe_mul(r, g[j]);
    return r;
  };
  Fr sm_l2 = Fr::zero(), sm_l3 = Fr::zero();   // s_m[ell + 2], s_m[ell + 3]
  for (int i = lane; i < n; i += 64) {
    const Fr s_i = svec(gam_i, i), s_i_inv = svec(gam_i_inv, i), s_m = svec(gam_m, i);
    const Fr u = buf[i < ell ? i : ell];
    // (1) CRS part over G | Hvec
    Fr k1 = fe_add(fe_mul(a2c, s_i), fe_mul(a3d, fe_mul(s_i_inv, u)));
    if (i < ell) k1 = fe_add(k1, a1b);
    if (i < ell + 2) k1 = fe_add(k1, fe_mul(a4x, s_m));   // G_b = G | Hvec[0..2) | G_t | G_u
    out_crs[i] = bad ? Fr::zero() : fe_neg(k1);
    // (2) instance part over R | S | T | U
    if (i < ell) {
      const Fr va = d.veca[(size_t)v.p * ell + i];
      out[i] = bad ? Fr::zero() : fe_neg(fe_mul(a7, va));
      out[ell + i] = bad ? Fr::zero() : fe_neg(fe_mul(a8, va));
      out[2 * ell + i] = bad ? Fr::zero() : fe_neg(fe_mul(a5x, s_m));
      out[3 * ell + i] = bad ? Fr::zero() : fe_neg(fe_mul(a6x, s_m));
    }
    if (i == ell + 2) sm_l2 = s_m;
    if (i == ell + 3) sm_l3 = s_m;
  }
  // s_m at the two positions the misc part needs: owned by the lanes (ell + 2) % 64 and (ell + 3) % 64
  {
    Fr t2 = sm_l2, t3 = sm_l3;
    CPX_UNROLL for (int j = 0; j < 8; j++) {
      t2.v[j] = (uint32_t)__shfl((int)sm_l2.v[j], (ell + 2) & 63, 64);
      t3.v[j] = (uint32_t)__shfl((int)sm_l3.v[j], (ell + 3) & 63, 64);
    }
    sm_l2 = t2;
    sm_l3 = t3;
  }
  // (3) misc part: singles, M, proof points.  Slot s sits at misc index s (SL_A == SL_M + 1).
  Fr w1, w2, w3, w4;   // weights of the four SameScalar equalities (same_scalar_argument.rs:127-137; include/cpx.h)
  if (d.rand_stride >= 12) {
    w1 = rnd[8];
    w2 = rnd[9];
    w3 = rnd[10];
    w4 = rnd[11];
  } else {
    w1 = fe_mul(a1, a2);
    w2 = fe_mul(a3, a4);
    w3 = fe_mul(a5, a6);
    w4 = fe_mul(a7, a8);
  }
  const Fr a4am = fe_mul(a4, alpha_m), a5am = fe_mul(a5, alpha_m), a6am = fe_mul(a6, alpha_m), a3ai = fe_mul(a3, alpha_i);
  for (int s = lane; s < d.NM; s += 64) {
    Fr k = Fr::zero();
    if (s == SL_H) {
      k = fe_sub(fe_sub(fe_sub(fe_mul(a2, fe_mul(fe_mul(fe_mul(alpha_i, alpha_i), z_ip), beta_i)), fe_mul(fe_mul(a2c, d_fin), beta_i)), fe_mul(a5x, sm_l2)),
                 fe_mul(a6x, sm_l3));
      k = fe_sub(fe_sub(k, fe_mul(w2, z_t)), fe_mul(w4, z_u));
    } else if (s == SL_GT) {
      k = fe_neg(fe_add(fe_mul(a4x, sm_l2), fe_mul(w1, z_t)));
    } else if (s == SL_GU) {
      k = fe_neg(fe_add(fe_mul(a4x, sm_l3), fe_mul(w3, z_u)));
    } else if (s == SL_GSUM) {
      k = fe_neg(fe_mul(a3ai, beta_g_inv));
    } else if (s == SL_HSUM) {
      k = fe_mul(a3ai, alpha_g);
    } else if (s == SL_M) {
      k = fe_neg(fe_mul(a1, alpha_sp));
    } else if (s == SL_A) {
      k = fe_sub(a4am, a1);
    } else if (s == SL_CMT1) {
      k = fe_add(a4am, fe_mul(w1, alpha_s));
    } else if (s == SL_CMT2) {
      k = fe_add(a5am, fe_mul(w2, alpha_s));
    } else if (s == SL_CMU1) {
      k = fe_add(a4am, fe_mul(w3, alpha_s));
    } else if (s == SL_CMU2) {
      k = fe_add(a6am, fe_mul(w4, alpha_s));
    } else if (s == SL_R) {
      k = fe_sub(a7, fe_mul(w2, z_k));
    } else if (s == SL_S) {
      k = fe_sub(a8, fe_mul(w4, z_k));
    } else if (s == SL_B) {
      k = fe_add(a1, a3ai);
    } else if (s == SL_C) {
      k = fe_mul(a2, alpha_i);
    } else if (s == SL_BC) {
      k = a2;
    } else if (s == SL_BD) {
      k = a3;
    } else if (s < sm.CMA1()) {   // IPA cross terms: L_C | R_C | L_D | R_D, L each
      const int q = s - SL_IPA0, blk = q / L, j = q % L;
      const Fr g = (blk & 1) ? gam_i_inv[j] : gam_i[j];
      k = fe_mul(blk < 2 ? a2 : a3, g);
    } else if (s == sm.CMA1()) {
      k = w1;
    } else if (s == sm.CMA2()) {
      k = w2;
    } else if (s == sm.CMB1()) {
      k = w3;
    } else if (s == sm.CMB2()) {
      k = w4;
    } else if (s == sm.BA()) {
      k = a4;
    } else if (s == sm.BT()) {
      k = a5;
    } else if (s == sm.BU()) {
      k = a6;
    } else {   // SameMSM cross terms: L_A | L_T | L_U | R_A | R_T | R_U, L each
      const int q = s - sm.LA(0), blk = q / L, j = q % L;
      const Fr g = blk < 3 ? gam_m[j] : gam_m_inv[j];
      const int col = blk % 3;
      k = fe_mul(col == 0 ? a4 : (col == 1 ? a5 : a6), g);
    }
    out[NI + s] = bad ? Fr::zero() : k;
  }
}

// fused batch: sum over the proofs of the CRS scalars (the shared bases G | Hvec merge in Fr; msm_accumulator.rs:47-51)
__global__ __launch_bounds__(64) void k_vs_crs_sum(const Fr* __restrict__ scal_crs, int B, int n, Fr* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Fr t = Fr::zero();
  for (int p = 0; p < B; p++) t = fe_add(t, scal_crs[(size_t)p * n + i]);
  out[i] = t;
}

// ------------------------------------------------------------------ launchers
static inline size_t lds_vec(const ProveDev& d) { return (size_t)d.n * 32 + 64; }
#define PS_LAUNCH(kern, B, lds, s, ...) hipLaunchKernelGGL(kern, dim3(B), dim3(64), lds, s, __VA_ARGS__)
void launch_ps_aperm(const ProveDev& d, int B, hipStream_t s) { PS_LAUNCH(k_ps_aperm, B, 0, s, d); }
void launch_ps_sameperm(const ProveDev& d, int B, hipStream_t s) { PS_LAUNCH(k_ps_sameperm, B, lds_vec(d), s, d); }
void launch_ps_gprod(const ProveDev& d, int B, hipStream_t s) { PS_LAUNCH(k_ps_gprod, B, lds_vec(d), s, d); }
void launch_ps_ipa_setup(const ProveDev& d, int B, hipStream_t s) { PS_LAUNCH(k_ps_ipa_setup, B, 64, s, d); }
void launch_ps_ipa_round(const ProveDev& d, int B, int j, hipStream_t s) { PS_LAUNCH(k_ps_ipa_round, B, 64, s, d, j); }
void launch_ps_smsm_setup(const ProveDev& d, int B, hipStream_t s) { PS_LAUNCH(k_ps_smsm_setup, B, 64 + 4 * 48, s, d); }
void launch_ps_smsm_round(const ProveDev& d, int B, int j, hipStream_t s) { PS_LAUNCH(k_ps_smsm_round, B, 64, s, d, j); }
void launch_ps_serialize(const ProveDev& d, int B, hipStream_t s) { PS_LAUNCH(k_ps_serialize, B, 0, s, d); }
void launch_vs_prefix(const VerifyDev& d, int B, hipStream_t s) { PS_LAUNCH(k_vs_prefix, B, (size_t)d.n * 32 + 64, s, d); }
void launch_vs_scalars(const VerifyDev& d, int B, hipStream_t s) { PS_LAUNCH(k_vs_scalars, B, (size_t)d.n * 32 + (size_t)4 * d.L * 32 + 64 + 4 * 48, s, d); }
void launch_vs_crs_sum(const Fr* d_scal_crs, int B, int n, Fr* d_out, hipStream_t s) {
  hipLaunchKernelGGL(k_vs_crs_sum, dim3((n + 63) / 64), dim3(64), 0, s, d_scal_crs, B, n, d_out);
}

}  // namespace cpx
