// Late log rounds of the batch prover on explicitly folded bases — gfx950 kernels, product code.
//
// The all-MSM prover pays n/2 (scalar, point) pairs per cross term in EVERY round of the inner-product and
// same-multiscalar arguments (inner_product_argument.rs:150-186, same_multiscalar_argument.rs:99-136), although the
// reference's folded bases shrink to n / 2^j points.  For the last log2(m) rounds this file goes back to folded bases (m = 16: four rounds,
// the shape of n <= 256; m = 32: five rounds, the default from n = 512 on, where an all-MSM round costs 2 - 4 times as much; m = 64: six
// rounds, option late_m — measured no better than 32):
//
//   1. materialise   G^(j0)_i = sum_t S_t G_(t m + i),  i < m,  for the five basis families G, G' = u o G, G_b (SameMSM CRS
//      basis), T_b, U_b — the identity the all-MSM form rests on, evaluated once at round j0 = log2(n) - log2(m):
//        k_late_fix      CRS families from the fixed-base table of multiples: one LANE per output walks its n/m points x 16
//                        windows = exactly (n/m) * 16 mixed additions, no buckets, no raw sets, no reduction kernels;
//        k_late_uniform  T_b, U_b (and G_b) from the 32-copy shifted tables: the fold coefficients S_t are the same for every
//                        output i and for T, U, G_b, so ALL lanes of a wave (lane = output) share one digit sequence: the
//                        (t, window) entries are counting-sorted by |digit| once per proof and every lane runs the same
//                        running-sum schedule  sum_b b B_b = sum_b (sum_{b' >= b} B_b')  without bucket storage:
//                        32 n/m mixed additions + 128 full additions per output.
//   2. k_late_tables   small multiples 1..8 of every materialised point and their endomorphism images, affine (one inversion per
//                      work-group): what a 4-bit signed window method needs.
//   3. k_late_msm      every cross term of a late round is an m/2-point MSM over the materialised points: Straus interleaving over
//                      the endomorphism-split scalars, 32 windows x (4 doublings + m additions), the windows spread over a few
//                      lanes per output; k_msm_tail joins the window groups (+ the round's single CRS term, k_late_fix).
// Same group elements as the all-MSM form, hence the same proof bytes (tests: option late_rounds = 0 / 4 against the oracle
// and the reference KAT).  Work per proof at ell = 252: ~58 k mixed-addition equivalents instead of 115 k for rounds 4..7.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <stdexcept>
#include "g1.hpp"
#include "g1_28.hpp"
#include "glv.hpp"
#include "recode.hpp"
#include "kernels.h"
#include "block_inverse.hpp"

namespace cpx {

namespace {
// acc += s * P_col over the table of multiples: 256 / CB mixed additions, digits recoded on the fly (signed radix-2^CB digits
// exactly as k_msm_fix recodes them: carry into the next window when a chunk reaches 2^(CB-1)).  The scalar sits in a shift
// register, so the loop stays rolled with static register indices: ONE copy of the addition per call site.
template <int CB, bool INL> __device__ __forceinline__ TAcc late_fix_accumulate(TAcc acc, const Fr& mont, const TFix* __restrict__ tab, int nc, uint32_t col) {
  constexpr int W = FixWin<CB>::W;
  constexpr uint32_t M = 1u << (CB - 1);
  const Fr kc = fe_from_mont(mont);
  FixDigitStream<CB> ds(kc.v);   // recode.hpp
  auto next_digit = [&]() { return ds.next(); };
  auto fetch = [&](int w, int d) { return d ? tab[((size_t)w * M + (size_t)((d < 0 ? -d : d) - 1)) * nc + col].a : TAff::identity(); };
  int dn = next_digit();
  TAff pn = fetch(0, dn);
#pragma unroll 1
  for (int w = 0; w < W; w++) {
    const int d = dn;
    const TAff p = pn;
    dn = w + 1 < W ? next_digit() : 0;
    pn = fetch(w + 1, dn);   // the next entry is in flight while this one is added
    if (d) acc = INL ? t_acc_add_mixed_inl(acc, t_cneg_lazy(p, d < 0)) : t_acc_add_mixed(acc, t_cneg(p, d < 0));
  }
  return acc;
}
}  // namespace

// ------------------------------------------------------------------ 1a. CRS families from the table of multiples
// lane g = p * m + i:  out[p * ostride + i] = sum_{t < n/m} scal[p * sstride + t m + i] * P_col(t m + i),  col = cols ? cols[k] : k
template <int CB> __global__ __launch_bounds__(64, 2) void k_late_fix(const Fr* __restrict__ scal, size_t sstride, const uint32_t* __restrict__ cols, int n, int m,
                                                                      const TFix* __restrict__ tab, int nc, TJac* __restrict__ out, size_t ostride, int total) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= total) return;
  const int p = g / m, i = g % m;
  const Fr* s = scal + (size_t)p * sstride;
  TAcc acc = TAcc::identity();
#pragma unroll 1
  for (int k = i; k < n; k += m) acc = late_fix_accumulate<CB, true>(acc, s[k], tab, nc, cols ? cols[k] : (uint32_t)k);
  out[(size_t)p * ostride + i] = t_acc_to_jac(acc);
}

// ------------------------------------------------------------------ 1b. T_b, U_b, G_b from the 32-copy shifted tables
// `waves` waves per proof (one while nfam * m * split <= 64: m = 16 with two lanes per output, m = 32 with one; m = 64: two waves, each
// sorting the proof's digit list for itself).  lane = family * m + i (family 0: T_b, 1: U_b, 2: G_b via the CRS copies); the n/m fold coefficients
// S_t = sm[t m] are split (glv.hpp) and recoded once, their 32 n/m (t, copy) entries sorted by |digit| into `list`
// (descending), and every lane adds ITS point of each entry, folding the running sum into its total once per magnitude.
constexpr int LU_MAX_T = 128;                       // n / m <= 128
__global__ __launch_bounds__(64, 2) void k_late_uniform(const Fr* __restrict__ sm, size_t sstride, const TAff* __restrict__ ptab, size_t ptab_proof_stride, int NP,
                                                        const TAff* __restrict__ ctab, int nc, const uint32_t* __restrict__ gb_cols, int n, int m, int nfam,
                                                        int split, int waves, TJac* __restrict__ out, size_t fstride) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int nt = n / m, nent = 32 * nt;
  TAcc* total = reinterpret_cast<TAcc*>(smem);                                    // [64]
  uint16_t* list = reinterpret_cast<uint16_t*>(smem + 64 * sizeof(TAcc));         // [nent] (t << 6) | (copy << 1) | neg
  uint32_t* cnt = reinterpret_cast<uint32_t*>(list + ((nent + 1) & ~1));          // [129] entries per magnitude
  uint32_t* cur = cnt + 132;                                                      // [129] scatter cursors
  const int p = blockIdx.x / waves, lane = threadIdx.x, flat = (blockIdx.x % waves) * 64 + lane;   // flat: the lane's index among the proof's lanes
  for (int b = lane; b < 132; b += 64) cnt[b] = 0;
  __syncthreads();
  // digits of S_t (lane t; nt may exceed 64)
  for (int t0 = 0; t0 < nt; t0 += 64) {
    const int t = t0 + lane;
    if (t < nt) {
      const Fr k = fe_from_mont(sm[(size_t)p * sstride + (size_t)t * m]);
      uint32_t tt[4], qq[4], bytes[8], nk, ntg;
      glv_split(k.v, tt, qq, nk, ntg);
      glv_biased_bytes(tt, bytes);
      glv_biased_bytes(qq, bytes + 4);
      CPX_UNROLL for (int c = 0; c < 32; c++) {
        const int d = (int)((bytes[c >> 2] >> (8 * (c & 3))) & 255u) - 128;
        if (d) atomicAdd(&cnt[d < 0 ? -d : d], 1u);
      }
    }
  }
  __syncthreads();
  if (lane == 0) {   // offsets in descending magnitude order
    uint32_t o = 0;
    for (int b = 128; b >= 1; b--) {
      cur[b] = o;
      o += cnt[b];
    }
  }
  __syncthreads();
  for (int t0 = 0; t0 < nt; t0 += 64) {
    const int t = t0 + lane;
    if (t < nt) {
      const Fr k = fe_from_mont(sm[(size_t)p * sstride + (size_t)t * m]);
      uint32_t tt[4], qq[4], bytes[8], nk, ntg;
      glv_split(k.v, tt, qq, nk, ntg);
      glv_biased_bytes(tt, bytes);
      glv_biased_bytes(qq, bytes + 4);
      const uint32_t sg_lo = nk ^ ntg, sg_hi = nk;
      CPX_UNROLL for (int c = 0; c < 32; c++) {
        const int d = (int)((bytes[c >> 2] >> (8 * (c & 3))) & 255u) - 128;
        if (d) {
          const uint32_t neg = (d < 0 ? 1u : 0u) ^ (c < 16 ? sg_lo : sg_hi);
          list[atomicAdd(&cur[d < 0 ? -d : d], 1u)] = (uint16_t)(((uint32_t)t << 6) | ((uint32_t)c << 1) | neg);
        }
      }
    }
  }
  __syncthreads();
  // lane = (sub * nfam + fam) * m + i: `split` lanes share an output, lane `sub` taking every split-th entry of every magnitude level
  // with a running sum and a total of its own (sum_b b B_b is linear in the buckets); the totals are added at the end.  With two
  // families (T_b, U_b: G_b then comes from the table of multiples) and split = 2 all 64 lanes work: 256 + 128 additions deep
  // instead of 512 + 128 on 48 lanes.
  // (split > 1 only with one wave per proof: the lanes of an output then meet in this wave's LDS)
  const int fam = (flat / m) % nfam, sub = flat / (m * nfam), i = flat % m;
  const bool live = sub < split;
  // where this lane's copy-0 points live and how far apart the copies are
  const TAff* base = nullptr;
  size_t cstride = 0;
  if (live) {
    if (fam < 2) {
      base = ptab + (size_t)p * ptab_proof_stride + 1 + (size_t)fam * n;   // row layout M | T_b | U_b
      cstride = (size_t)NP;
    } else {
      base = ctab;
      cstride = (size_t)nc;
    }
  }
  TAcc acc = TAcc::identity();
  total[lane] = TAcc::identity();
  uint32_t pos = 0;
  for (int b = 128; b >= 1; b--) {
    const uint32_t c = cnt[b];
    if (live) {
      for (uint32_t e = pos + (uint32_t)sub; e < pos + c; e += (uint32_t)split) {
        const uint32_t en = list[e];
        const uint32_t k = (en >> 6) * (uint32_t)m + (uint32_t)i;
        const TAff q = base[(size_t)((en >> 1) & 31u) * cstride + (fam < 2 ? k : gb_cols[k])];
        acc = t_acc_add_mixed_inl(acc, t_cneg_lazy(q, (en & 1u) != 0));
      }
      if (pos + c) total[lane] = t_acc_add(total[lane], acc);   // the running sum waits in LDS between its additions (VGPR budget)
    }
    pos += c;
  }
  __syncthreads();
  if (live && sub == 0) {
    TAcc sum = total[lane];
    for (int s2 = 1; s2 < split; s2++) sum = t_acc_add(sum, total[lane + s2 * nfam * m]);
    out[(size_t)fam * fstride + (size_t)p * m + i] = t_acc_to_jac(sum);
  }
}

// ------------------------------------------------------------------ 2. small multiples of the materialised points
// lane per point g (of `total`): jac[g] -> tab[g][16] = { j P : j = 1..8 } | { j (z^2 P) = j (-phi(P)) : j = 1..8 }, affine.
// The eight Jacobian multiples wait in `tmp` (8 TblTmp per lane) for the work-group's one inversion.
constexpr int LATE_MULT = 8;
__global__ __launch_bounds__(64) void k_late_tables(const TJac* __restrict__ jac, TAff* __restrict__ tab, TblTmp* __restrict__ tmp, int total) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  TF* buf = reinterpret_cast<TF*>(smem);
  const int g = blockIdx.x * 64 + threadIdx.x;
  const bool live = g < total;
  const TJac P = live ? jac[g] : TJac::identity();
  const bool inf = P.is_identity();
  TblTmp* my = tmp + (size_t)(blockIdx.x * 64 + threadIdx.x) * LATE_MULT;
  TF prod = t_one();
  if (live && !inf) {
    // 1P, 2P = dbl, 3P = 2P + P, 4P = dbl(2P), 5P = 4P + P, 6P = dbl(3P), 7P = 6P + P, 8P = dbl(4P)
    for (int j = 0; j < LATE_MULT; j++) {
      TJac q;
      if (j == 0) q = P;
      else if (j & 1) q = t_dbl(my[j / 2].p);          // (j + 1) even: 2 * ((j + 1) / 2) P, stored at index (j + 1) / 2 - 1 = j / 2
      else q = t_add(my[j - 1].p, P);
      my[j].p = q;
      my[j].pre = prod;
      prod = t_mul(prod, q.z);   // multiples of a point of prime order r > 8 are never the identity
    }
  }
  TF inv = t_block_batch_inverse(prod, buf);
  if (!live) return;
  TAff* o = tab + (size_t)g * (2 * LATE_MULT);
  const TF beta = t_beta();
  for (int j = LATE_MULT - 1; j >= 0; j--) {
    TAff a = TAff::identity(), e = TAff::identity();
    if (!inf) {
      const TJac q = my[j].p;
      const TF zinv = t_mul(inv, my[j].pre);
      inv = t_mul(inv, q.z);
      a = t_to_affine(q, zinv);
      e = TAff{t_mul(a.x, beta), t_neg(a.y)};   // z^2 (x, y) = -phi(x, y) = (beta x, -y)  (glv.hpp)
    }
    o[j] = a;
    o[LATE_MULT + j] = e;
  }
}

// ------------------------------------------------------------------ 3. the cross terms of a late round
// Output (p, o) = sum_{s < hm} scal[p][o.scal_off + s] * F^(j0)_(pt(s)),  pt(s) = the s-th index below m whose bit `half` is set
// (o.hi) or clear — the index pattern of k_ipa_round_scalars / k_smsm_round_scalars, which produce the scalars.
// Straus interleaving over the endomorphism-split scalars with 4-bit signed digits (32 windows per half), the windows of an
// output spread over `slices` LANES (a lone lane per output is a chain of 128 doublings + 16 x 32 additions: at 4 .. 6 outputs
// per proof the GPU would stand three quarters empty for its whole length): lane g = (p * nout + o) * slices + slice takes the
// windows [slice * 32 / slices, (slice + 1) * 32 / slices) and leaves the partial sum out[g] of weight 2^(128 / slices * slice);
// launch_msm_tail(group = slices, shift = 128 / slices) adds them up (and the round's single CRS term as an extra).
__global__ __launch_bounds__(64, 2) void k_late_msm(const LateRound r) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* dw = reinterpret_cast<uint32_t*>(smem);   // [2 hm * wph][64]: the lane's digit nibbles (biased by 7), word-interleaved by lane
  const int lane = threadIdx.x, g = blockIdx.x * 64 + lane;
  const int hm = r.m / 2, ws = 32 / r.slices, wph = (ws + 7) / 8;   // windows per slice, digit words per scalar half
  if (g >= r.total) return;   // (no barrier below: a lane reads back only what it wrote)
  const int og = g / r.slices, slice = g % r.slices;
  const int p = og / r.nout, oi = og % r.nout;
  const LateOut o = r.o[oi];
  const Fr* sc = r.scal + (size_t)p * r.scal_proof_stride + o.scal_off;
  const int w_lo = slice * ws;
  uint64_t signs = 0;   // bit 2 s: sign of the |t| half of scalar s, bit 2 s + 1: of the q half (s < m / 2 <= 32)
  for (int s = 0; s < hm; s++) {
    const Fr k = fe_from_mont(sc[s]);
    uint32_t h[2][4], nk, ntg;
    glv_split(k.v, h[0], h[1], nk, ntg);
    signs |= ((uint64_t)(nk ^ ntg) << (2 * s)) | ((uint64_t)nk << (2 * s + 1));
    CPX_UNROLL for (int hf = 0; hf < 2; hf++) {
      uint32_t packed[4];
      recode_signed_nibbles_biased(h[hf], packed);   // recode.hpp: signed radix-16 digits in [-7, 8], biased by 7, eight per word
      // keep the words that hold this slice's windows
      for (int x = 0; x < wph; x++) {
        uint32_t word = 0;
        CPX_UNROLL for (int wd = 0; wd < 4; wd++)
          if (wd == (w_lo >> 3) + x) word = packed[wd];
        dw[(size_t)((s * 2 + hf) * wph + x) * 64 + lane] = word;
      }
    }
  }
  const TAff* ftab = r.tab + ((size_t)o.fam * r.nproofs + p) * (size_t)r.m * (2 * LATE_MULT);
  TAcc acc = TAcc::identity();
#pragma unroll 1
  for (int win = w_lo + ws - 1; win >= w_lo; win--) {
    if (win != w_lo + ws - 1) {
#pragma unroll 1
      for (int q = 0; q < 4; q++) acc = t_acc_dbl(acc);
    }
    const int rel = win - (w_lo & ~7);   // nibble index inside the kept words
#pragma unroll 1
    for (int sh = 0; sh < 2 * hm; sh++) {   // (scalar s, half hf) flattened: one copy of the inlined addition
      const int s = sh >> 1, hf = sh & 1;
      const int pt = (s / r.half) * 2 * r.half + (s % r.half) + (o.hi ? r.half : 0);
      const uint32_t word = dw[(size_t)(sh * wph + (rel >> 3)) * 64 + lane];
      const int d = (int)((word >> (4 * (rel & 7))) & 15u) - 7;
      if (d) {
        const bool neg = (d < 0) != (((signs >> sh) & 1u) != 0u);
        acc = t_acc_add_mixed_inl(acc, t_cneg_lazy(ftab[(size_t)pt * (2 * LATE_MULT) + hf * LATE_MULT + (d < 0 ? -d : d) - 1], neg));
      }
    }
  }
  r.out[g] = t_acc_to_jac(acc);
}

// SG'[k] = 1 for k < m: the rescaled basis G' = u o G has been materialised with its weights, the late rounds restart from unit
// fold coefficients (vec: [B][4][n] = c | d | SG | SGp, kernels.hip "Fr algebra of the log rounds")
__global__ __launch_bounds__(64) void k_late_restart(Fr* __restrict__ vec, int n, int m, int total) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= total) return;
  vec[(size_t)(g / m) * 4 * n + 3 * (size_t)n + (g % m)] = Fr::one();
}

// ------------------------------------------------------------------ launchers
#define LATE_LAUNCH(kern, grid, block, lds, stream, ...)                                              \
  do {                                                                                                \
    hipEvent_t _a = nullptr, _b = nullptr;                                                            \
    take_launch_events(&_a, &_b);                                                                     \
    if (_a || _b) hipExtLaunchKernelGGL(kern, grid, block, lds, stream, _a, _b, 0, __VA_ARGS__);      \
    else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                             \
  } while (0)

size_t late_tab_entries() { return 2 * LATE_MULT; }
size_t late_tmp_per_lane() { return LATE_MULT; }
// (m a power of two up to 64: k_late_msm keeps one sign bit per endomorphism half of its m / 2 scalars in a 64-bit mask)
bool late_supported(int n, int m) { return m >= 2 && m <= 64 && (m & (m - 1)) == 0 && n % m == 0 && n / m <= LU_MAX_T && n > m; }

void launch_late_fix(const Fr* d_scal, size_t sstride, const uint32_t* d_cols, int n, int m, const TFix* d_fix_tab, int cbits, int nc, TJac* d_out, size_t ostride,
                     int nproofs, hipStream_t s) {
  const int total = nproofs * m;
  if (total <= 0) return;
  if (cbits == 19) LATE_LAUNCH(k_late_fix<19>, dim3((total + 63) / 64), dim3(64), 0, s, d_scal, sstride, d_cols, n, m, d_fix_tab, nc, d_out, ostride, total);
  else if (cbits == 16) LATE_LAUNCH(k_late_fix<16>, dim3((total + 63) / 64), dim3(64), 0, s, d_scal, sstride, d_cols, n, m, d_fix_tab, nc, d_out, ostride, total);
  else LATE_LAUNCH(k_late_fix<8>, dim3((total + 63) / 64), dim3(64), 0, s, d_scal, sstride, d_cols, n, m, d_fix_tab, nc, d_out, ostride, total);
}
void launch_late_uniform(const Fr* d_sm, size_t sstride, const TAff* d_ptab, size_t ptab_proof_stride, int NP, const TAff* d_ctab, int nc, const uint32_t* d_gb_cols, int n,
                         int m, int nfam, int split, TJac* d_out, size_t fstride, int nproofs, hipStream_t s) {
  if (nproofs <= 0) return;
  // (late_supported() keeps the engine away from shapes this kernel cannot take; a caller that gets here anyway must not go on with
  // uninitialised folded bases)
  const int waves = (nfam * m * split + 63) / 64;
  if ((split > 1 && waves > 1) || split < 1 || m <= 0 || n % m || n / m > LU_MAX_T || (n / m) > (1 << 10))
    throw std::invalid_argument("launch_late_uniform: the lanes of a split output must fit one wave and n / m the 16-bit list entries");
  const int nent = 32 * (n / m);
  const size_t lds = 64 * sizeof(TAcc) + (size_t)((nent + 1) & ~1) * 2 + 2 * 132 * 4;
  LATE_LAUNCH(k_late_uniform, dim3(nproofs * waves), dim3(64), lds, s, d_sm, sstride, d_ptab, ptab_proof_stride, NP, d_ctab, nc, d_gb_cols, n, m, nfam, split, waves, d_out,
              fstride);
}
void launch_late_tables(const TJac* d_jac, TAff* d_tab, TblTmp* d_tmp, int npoints, hipStream_t s) {
  if (npoints <= 0) return;
  LATE_LAUNCH(k_late_tables, dim3((npoints + 63) / 64), dim3(64), 2 * 64 * sizeof(TF), s, d_jac, d_tab, d_tmp, npoints);
}
void launch_late_msm(const LateRound& r, hipStream_t s) {
  if (r.total <= 0) return;
  const int ws = 32 / r.slices, wph = (ws + 7) / 8;
  const size_t lds = (size_t)r.m * wph * 64 * 4;
  LATE_LAUNCH(k_late_msm, dim3((r.total + 63) / 64), dim3(64), lds, s, r);
}
void launch_late_restart(Fr* d_vec, int n, int m, int nproofs, hipStream_t s) {
  const int total = nproofs * m;
  if (total > 0) LATE_LAUNCH(k_late_restart, dim3((total + 63) / 64), dim3(64), 0, s, d_vec, n, m, total);
}

}  // namespace cpx
