// Fused log rounds of the device-resident batch prover (gfx950) — product code.
//
// One launch per round of the inner-product argument and of the SameMSM argument
// (/root/reference/src/inner_product_argument.rs:150-186, same_multiscalar_argument.rs:99-136) for small and mid-size batches, where a
// round used to be a chain of six to ten latency-bound launches (scalars -> k_msm_fix [-> k_msm_tblw] -> k_reduce_sets_wave ->
// k_finalize_ranges -> k_ps_*_round -> fold): every kernel of that chain waited for the slowest wave of the one before, over the whole batch.
//
//   grid = proofs x waves-per-proof single-wave work-groups.  A wave is an MSM wave of ONE cross term of ONE proof — the fixed-base body
//   (CRS bases from the table of multiples) or the bucket-list body (per-proof shifted tables), msm_body.hpp, unchanged — that
//     1. accumulates its share of the cross term,
//     2. reduces its own 64 (or 2 x 64) lane accumulators with the quad-cooperative programs of k_reduce_sets_wave, in its own LDS
//        (the SIMD it sits on has nothing else to do when the batch is small),
//     3. leaves ONE partial sum per set in global memory and counts itself in on the proof's arrival counter.
//   The LAST wave of a proof to arrive runs the rest of the round for that proof alone ("tail"): the tree over the partial sums of each
//   cross term, one shared inversion for the 4 / 6 results, their 48-byte encodings into the slot registry, the transcript step
//   (wave_strobe.hpp), the challenge's inverse, the folds of the scalar vectors and the scalars of the NEXT round's cross terms.
//   Proofs do not wait for each other inside a round; a round is one dependent launch instead of six to ten.
//
// Same group elements and the same transcript as the unfused path (tests/test_gpu_parity.py::test_engine_variants_stay_bit_exact runs
// both on the reference's KAT).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdexcept>
#include "mont32.hpp"
#include "modinv30.hpp"
#include "strobe.hpp"
#include "wave_strobe.hpp"
#include "protocol.h"
#include "block_inverse.hpp"
#include "msm_body.hpp"

namespace cpx {

#define LBL(s) s, (sizeof(s) - 1)

namespace {

// ---- Fr helpers of the tail (out of line: sixteen dwords of arguments travel in registers) ----
__device__ __noinline__ Fr frm(const Fr& a, const Fr& b) { return fe_mul(a, b); }
__device__ __noinline__ Fr fr_inv_ool(const Fr& a) { return fr_inv_divsteps(a); }
__device__ __forceinline__ Fr shfl_xor_fr(const Fr& v, int mask) {
  Fr r;
  CPX_UNROLL for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl_xor((int)v.v[i], mask, 64);
  return r;
}
__device__ Fr wave_sum_fr(Fr v) {
  for (int m = 32; m >= 1; m >>= 1) v = fe_add(v, shfl_xor_fr(v, m));
  return v;
}
__device__ __forceinline__ void tail_sync() {   // one wave: orders its lanes' global / LDS accesses
  __threadfence_block();
  __syncthreads();
}

// ---- the Fr side of a round on the 64 lanes of the tail wave (the arithmetic of k_ipa_round_fold / _scalars, k_smsm_round_*) ----
// IPA.  vec: c | d | SG | SGp, n elements each.  c_L += gamma^-1 c_R, d_L += gamma d_R, SG[hi] *= gamma, SGp[hi] *= gamma^-1
// (inner_product_argument.rs:174-184 on the scalar side of the all-MSM form)
__device__ void ipa_fold_wave(Fr* vec, int n, int half, const Fr& g, const Fr& gi) {
  const int lane = threadIdx.x, hn = n / 2;
  Fr *c = vec, *d = vec + n, *SG = vec + 2 * n, *SGp = vec + 3 * n;
  for (int i = lane; i < half; i += 64) {
    c[i] = fe_add(c[i], frm(gi, c[half + i]));
    d[i] = fe_add(d[i], frm(g, d[half + i]));
  }
  for (int t = lane; t < hn; t += 64) {
    const int kh = (t / half) * 2 * half + (t % half) + half;
    SG[kh] = frm(SG[kh], g);
    SGp[kh] = frm(SGp[kh], gi);
  }
}
// the scalars of the four cross terms of the round with this `half` (inner_product_argument.rs:150-163):
// out = [ L_C (n/2), beta <c_L, d_R> | L_D (n/2) | R_C (n/2), beta <c_R, d_L> | R_D (n/2) ]
__device__ void ipa_scalars_wave(const Fr* vec, int n, int half, const Fr& beta, Fr* o) {
  const int lane = threadIdx.x, hn = n / 2;
  const Fr *c = vec, *d = vec + n, *SG = vec + 2 * n, *SGp = vec + 3 * n;
  for (int t = lane; t < hn; t += 64) {
    const int kl = (t / half) * 2 * half + (t % half), kh = kl + half;
    const int ih = kh & (half - 1), il = kl & (half - 1);
    o[t] = frm(c[ih], SG[kh]);
    o[hn + 1 + t] = frm(d[half + il], SGp[kl]);
    o[2 * hn + 1 + t] = frm(c[half + il], SG[kl]);
    o[3 * hn + 2 + t] = frm(d[ih], SGp[kh]);
  }
  Fr ip1 = Fr::zero(), ip2 = Fr::zero();
  for (int i = lane; i < half; i += 64) {
    ip1 = fe_add(ip1, frm(c[i], d[half + i]));
    ip2 = fe_add(ip2, frm(c[half + i], d[i]));
  }
  ip1 = wave_sum_fr(ip1);
  ip2 = wave_sum_fr(ip2);
  if (lane == 0) {
    o[hn] = frm(beta, ip1);
    o[3 * hn + 1] = frm(beta, ip2);
  }
}
// SameMSM.  vec: x | SM.  x_L += gamma^-1 x_R, SM[hi] *= gamma (same_multiscalar_argument.rs:126-134)
__device__ void smsm_fold_wave(Fr* vec, int n, int half, const Fr& g, const Fr& gi) {
  const int lane = threadIdx.x, hn = n / 2;
  Fr *x = vec, *SM = vec + n;
  for (int i = lane; i < half; i += 64) x[i] = fe_add(x[i], frm(gi, x[half + i]));
  for (int t = lane; t < hn; t += 64) {
    const int kh = (t / half) * 2 * half + (t % half) + half;
    SM[kh] = frm(SM[kh], g);
  }
}
// out = [ L_* scalars (n/2) | R_* scalars (n/2) ]  (same_multiscalar_argument.rs:104-112)
__device__ void smsm_scalars_wave(const Fr* vec, int n, int half, Fr* o) {
  const int lane = threadIdx.x, hn = n / 2;
  const Fr *x = vec, *SM = vec + n;
  for (int t = lane; t < hn; t += 64) {
    const int kl = (t / half) * 2 * half + (t % half), kh = kl + half;
    o[t] = frm(x[kh & (half - 1)], SM[kh]);
    o[hn + t] = frm(x[(kl & (half - 1)) + half], SM[kl]);
  }
}

// ---- in-wave reduction of one raw set (the body of k_reduce_sets_wave) ----
__device__ __forceinline__ void reduce_set_inwave(const uint32_t* __restrict__ set, int prog_index, TAcc* buf, TAcc* __restrict__ out) {
  const QuadProgram& prog = c_quad_prog[prog_index];
  const int lane = threadIdx.x;
  buf[lane] = raw_load(set, lane);
  if (lane == 0) buf[QP_IDENT] = TAcc::identity();
  __syncthreads();
  quad_exec(buf, prog.nrounds, [&](int r, int quad, bool& dbl) {
    dbl = prog.dbl[r] != 0;
    return prog.job[r][quad];
  });
  if (lane == 0) *out = buf[prog.result];
  __syncthreads();
}

constexpr int TAIL_MAX_REQ = 6, TAIL_PER_REQ = 8;   // partial sums per request the tail's tree takes (fixed-base: 16 / wpw, tables: 64 / wpw)
constexpr int TAIL_IDENT = TAIL_MAX_REQ * TAIL_PER_REQ;
constexpr size_t TAIL_LDS = (TAIL_IDENT + 1) * sizeof(TAcc) + 2 * 64 * sizeof(TF) + 64;

// The rest of round j for proof p, on one wave.
template <bool IPA> __device__ void round_tail(const RoundDev& rd, const ProveDev& d, int p, int j, uint8_t* smem) {
  TAcc* buf = reinterpret_cast<TAcc*>(smem);                                             // [TAIL_IDENT + 1]
  TF* inv_buf = reinterpret_cast<TF*>(smem + (TAIL_IDENT + 1) * sizeof(TAcc));           // [2 * 64]
  uint8_t* scratch = smem + (TAIL_IDENT + 1) * sizeof(TAcc) + 2 * 64 * sizeof(TF);       // [64]
  const int lane = threadIdx.x, nreq = rd.nreq;
  const uint32_t* first = rd.meta + (size_t)p * nreq;
  const uint32_t* count = rd.meta + (size_t)rd.nproofs * nreq + (size_t)p * nreq;
  const uint32_t* cidx = rd.comp_index + (size_t)p * nreq;
  // 1. the partial sums of request r in entries [r * 8, r * 8 + 8), identity-padded; a tree over each octet, all requests side by side
  if (lane < nreq * TAIL_PER_REQ) {
    const int r = lane / TAIL_PER_REQ, i = lane % TAIL_PER_REQ;
    buf[lane] = (uint32_t)i < count[r] ? rd.part[first[r] + i] : TAcc::identity();
  }
  if (lane == 0) buf[TAIL_IDENT] = TAcc::identity();
  __syncthreads();
  {
    int sft = TAIL_PER_REQ / 2;
    while (sft > 1 && (uint32_t)sft >= rd.max_count) sft >>= 1;   // (uniform: the largest count of the round's requests)
    if ((uint32_t)sft >= rd.max_count) sft = 0;                   // one partial per request: nothing to add
    int rounds = 0;
    for (int s = sft; s >= 1; s >>= 1) rounds += (nreq * s + 15) / 16;
    int cur = sft, done = 0;
    quad_exec(buf, rounds, [&](int, int quad, bool&) {
      const int jb = done + quad, njobs = nreq * cur;
      QuadJob job{(uint8_t)TAIL_IDENT, (uint8_t)TAIL_IDENT, (uint8_t)TAIL_IDENT, 0};
      if (jb < njobs) {
        const int r = jb / cur, i = jb % cur, a = r * TAIL_PER_REQ + i;
        job = QuadJob{(uint8_t)a, (uint8_t)(a + cur), (uint8_t)a, 1};
      }
      done += 16;
      if (done >= njobs) {
        cur >>= 1;
        done = 0;
      }
      return job;
    });
  }
  // 2. lanes r < nreq: Jacobian, one shared inversion, affine, standard form, 48-byte encoding into the slot registry
  {
    TJac pt = TJac::identity();
    if (lane < nreq) pt = t_acc_to_jac(buf[lane * TAIL_PER_REQ]);
    const bool inf = pt.is_identity();
    const TF zinv = t_block_batch_inverse(pt.z, inv_buf);
    if (lane < nreq) {
      const Aff a = inf ? Aff::identity() : t_to_std(t_to_affine(pt, zinv));
      aff_compress_store(a, d.slotcomp + (size_t)cidx[lane] * 48);
    }
  }
  tail_sync();
  // 3. the transcript step: hash the round's cross terms, draw gamma (inner_product_argument.rs:164-171,
  //    same_multiscalar_argument.rs:114-123)
  uint64_t* tstate = d.tstate + (size_t)p * 27;
  WaveStrobe t;
  t.load(tstate, lane);
  for (int r = 0; r < nreq; r++) {
    if (IPA) t.append_message(LBL("ipa_loop"), d.slotcomp + (size_t)cidx[r] * 48, 48, scratch);
    else t.append_message(LBL("same_msm_loop"), d.slotcomp + (size_t)cidx[r] * 48, 48, scratch);
  }
  const Fr gamma = IPA ? t.challenge_scalar(LBL("ipa_gamma"), scratch) : t.challenge_scalar(LBL("same_msm_gamma"), scratch);
  t.store(tstate);
  const Fr gi = fr_inv_ool(gamma);
  if (lane == 0) {
    d.rgam[2 * (size_t)p] = gamma;
    d.rgam[2 * (size_t)p + 1] = gi;
  }
  // 4. the folds, then the scalars of the next round's cross terms
  const int n = d.n, half = n >> (j + 1);
  if (IPA) {
    Fr* vec = d.rvec + (size_t)p * 4 * n;
    ipa_fold_wave(vec, n, half, gamma, gi);
    if (half > 1 && rd.next_scalars) {
      tail_sync();
      ipa_scalars_wave(vec, n, half / 2, d.rbeta[p], rd.scal_out + (size_t)p * (2 * n + 2));
    }
  } else {
    Fr* vec = d.rvec2 + (size_t)p * 2 * n;
    smsm_fold_wave(vec, n, half, gamma, gi);
    if (half > 1 && rd.next_scalars) {
      tail_sync();
      smsm_scalars_wave(vec, n, half / 2, rd.scal_out + (size_t)p * n);
    }
  }
  if (lane == 0) rd.counter[p] = 0;   // ready for the next round's launch
}

}  // namespace

// FW: windows per fixed-base wave (16 / FW waves per task); TW: windows per table wave (32 / TW waves per task; 0: no table tasks).
// COMBINE (SameMSM rounds of a batch that does not fill the GPU): nf * (16 / FW) == nt * (32 / TW) and wave w of a proof runs fixed-base
// block w AND table block w, one after the other — as many waves per proof as SIMDs per proof (128 proofs: 8), none shares a SIMD with
// another (a lone chain beside a second wave takes 1.7 times as long, and the slowest wave is the round).
template <int FW, int TW, bool IPA, bool COMBINE = false> __global__ __launch_bounds__(64, 2) void k_round_fused(const RoundDev rd, const ProveDev d, int j) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t FWG = 16 / FW;
  const uint32_t bid = blockIdx.x, p = bid / rd.wpp, w = bid % rd.wpp;
  const uint32_t nfw = (uint32_t)rd.nf * FWG;
  const int lane = threadIdx.x;
  TAcc* buf = reinterpret_cast<TAcc*>(smem);
  if (COMBINE || w < nfw) {
    const uint32_t bf = p * nfw + w;
    msm_fix_body<16, FW>(rd.ftasks, rd.fixtab, rd.nc, rd.fraw, rd.fraw_slot, bf);
    tail_sync();
    const uint32_t slot = rd.ftasks[bf / FWG].out_first + bf % FWG;
    reduce_set_inwave(rd.fraw + (size_t)bf * RAW_SET_WORDS, 0, buf, rd.part + slot);
  }
  if constexpr (TW != 0) {
    if (COMBINE || w >= nfw) {
      constexpr uint32_t TWV = 32 / TW;
      const uint32_t ntw = (uint32_t)rd.nt * TWV, bt = p * ntw + (COMBINE ? w : w - nfw);
      msm_tblw_body<TW, false>(rd.ttasks, rd.traw, rd.traw_slot, 1, bt);
      tail_sync();
      const uint32_t slot = rd.ttasks[bt / TWV].pad + 2 * (bt % TWV);
      reduce_set_inwave(rd.traw + (size_t)bt * 2 * RAW_SET_WORDS, 1, buf, rd.part + slot);
      reduce_set_inwave(rd.traw + ((size_t)bt * 2 + 1) * RAW_SET_WORDS, 2, buf, rd.part + slot + 1);
    }
  }
  // arrival: the partial sums above are released before the count, the last arriver acquires them
  __threadfence();
  uint32_t old = 0;
  if (lane == 0) old = atomicAdd(&rd.counter[p], 1u);
  old = (uint32_t)__shfl((int)old, 0, 64);
  if (old != rd.wpp - 1) return;
  __threadfence();
  round_tail<IPA>(rd, d, (int)p, j, smem);
}

// ------------------------------------------------------------------ launcher
#define ROUND_LAUNCH(kern, grid, block, lds, stream, ...)                                             \
  do {                                                                                                \
    hipEvent_t _a = nullptr, _b = nullptr;                                                            \
    take_launch_events(&_a, &_b);                                                                     \
    if (_a || _b) hipExtLaunchKernelGGL(kern, grid, block, lds, stream, _a, _b, 0, __VA_ARGS__);      \
    else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                             \
  } while (0)

bool round_fused_supported(int fix_wpw, int tbl_wpw, bool ipa) {
  if (ipa) return tbl_wpw == 0 && (fix_wpw == 4 || fix_wpw == 8 || fix_wpw == 16);
  if ((fix_wpw == 4 && tbl_wpw == 16) || (fix_wpw == 8 && tbl_wpw == 32)) return true;   // (also the shapes of combined waves, RoundDev::combine)
  return (fix_wpw == 8 || fix_wpw == 16) && (tbl_wpw == 8 || tbl_wpw == 16 || tbl_wpw == 32);
}
size_t round_fused_lds(int tbl_wpw) {
  size_t lds = std::max<size_t>((size_t)QP_ENTRIES * sizeof(TAcc), 16 * FIX_CHUNK * 2);
  if (tbl_wpw) lds = std::max<size_t>(lds, TBW_LDS);
  return std::max(lds, TAIL_LDS);
}
void launch_round_fused(const RoundDev& rd, const ProveDev& d, int j, bool ipa, hipStream_t s) {
  if (rd.nproofs <= 0) return;
  if (!round_fused_supported(rd.fix_wpw, rd.tbl_wpw, ipa) || rd.nreq > TAIL_MAX_REQ || rd.max_count > (uint32_t)TAIL_PER_REQ)
    throw std::invalid_argument("launch_round_fused: unsupported wave shape");
  const dim3 grid((unsigned)rd.nproofs * rd.wpp), block(64);
  const size_t lds = round_fused_lds(rd.tbl_wpw);
  if (ipa) {
    if (rd.fix_wpw == 4) ROUND_LAUNCH((k_round_fused<4, 0, true>), grid, block, lds, s, rd, d, j);
    else if (rd.fix_wpw == 8) ROUND_LAUNCH((k_round_fused<8, 0, true>), grid, block, lds, s, rd, d, j);
    else ROUND_LAUNCH((k_round_fused<16, 0, true>), grid, block, lds, s, rd, d, j);
    return;
  }
#define CASE(F, T) \
  if (rd.fix_wpw == F && rd.tbl_wpw == T) { ROUND_LAUNCH((k_round_fused<F, T, false>), grid, block, lds, s, rd, d, j); return; }
  if (rd.combine) {   // wave w = fixed-base block w + table block w: 4 windows of a CRS task + 16 of a table task (8 waves per proof), or 8 + 32 (4 waves per proof)
    if (rd.nf * (16 / rd.fix_wpw) != rd.nt * (32 / rd.tbl_wpw)) throw std::invalid_argument("launch_round_fused: combined waves need as many fixed-base blocks as table blocks");
    if (rd.fix_wpw == 4 && rd.tbl_wpw == 16) ROUND_LAUNCH((k_round_fused<4, 16, false, true>), grid, block, lds, s, rd, d, j);
    else if (rd.fix_wpw == 8 && rd.tbl_wpw == 32) ROUND_LAUNCH((k_round_fused<8, 32, false, true>), grid, block, lds, s, rd, d, j);
    else throw std::invalid_argument("launch_round_fused: combined waves take 4 / 16 or 8 / 32 windows per wave");
    return;
  }
  CASE(8, 8) CASE(8, 16) CASE(8, 32) CASE(16, 8) CASE(16, 16) CASE(16, 32)
#undef CASE
  // (4 / 16 exists as combined waves only: a round that reaches this line would add no cross terms, draw no challenge and fold nothing)
  throw std::invalid_argument("launch_round_fused: no kernel for this wave shape without combined waves");
}

}  // namespace cpx
