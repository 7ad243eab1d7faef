// The MSM wave bodies of the gfx950 kernels and what they share with the in-wave reductions — device code included by kernels.hip (the
// stand-alone MSM / reduction kernels) and round.hip (the fused log-round kernels: MSM waves that reduce their own accumulators and
// whose last arriver runs the round's finalisation, transcript step and Fr folds).  Moved out of kernels.hip unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include "g1.hpp"
#include "g1_28.hpp"
#include "g1_28_quad.hpp"
#include "quad_program.hpp"
#include "recode.hpp"
#include "glv.hpp"
#include "kernels.h"

namespace cpx {

// compressed encoding of a point to a 16-byte-aligned destination (every compressed-point buffer of the engine is: 48-byte records
// in hipMalloc'ed arrays): three 16-byte stores instead of 48 single-byte ones
__device__ __forceinline__ void aff_compress_store(const Aff& a, uint8_t* out) {
  uint32_t w[12];
  aff_compress_words(a, w);
  uint4* o = reinterpret_cast<uint4*>(out);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
  o[2] = make_uint4(w[8], w[9], w[10], w[11]);
}

// ------------------------------------------------------------------ raw lane accumulators ("sets")
constexpr int ACC_WORDS = (int)(sizeof(TAcc) / 4);   // 56
constexpr int RAW_SET_WORDS = ACC_WORDS * 64;
__device__ __forceinline__ void raw_store(uint32_t* __restrict__ set, int lane, const TAcc& a) { reinterpret_cast<TAcc*>(set)[lane] = a; }
__device__ __forceinline__ TAcc raw_load(const uint32_t* __restrict__ set, int lane) { return reinterpret_cast<const TAcc*>(set)[lane]; }

// Small grids (a few proofs in flight: latency matters, not throughput): one WAVE per set and every point addition spread over the
// four lanes of a quad (g1_28_quad.hpp: 6 us instead of 18-20 us per dependent addition), 16 additions per wave and round.  The
// rounds of a set are a fixed PROGRAM of (operand, operand, destination) jobs over an array of points in LDS, one job per quad:
//   plain    (k_msm_fix):  the tree over 64 lanes, 32 + 16 + ... + 1 additions in 7 rounds.
//   weighted (k_msm_tblw): sum_l (l + 1) A_l with l = 8 a + b is 8 sum_a a R_a + sum_b (b + 1) C_b over the row sums R_a and the
//            column sums C_b of the 8 x 8 arrangement (three tree levels for both, 7 rounds); a weighted sum of EIGHT points is,
//            bit by bit of the weights, three sums of four points (2 rounds), so that seven points with weights 2^0 .. 2^6 remain
//            (2^6: the plain sum, for the upper half of the magnitudes): shifted side by side (<= 6 doubling rounds) and added in a
//            tree (3 rounds).  13 addition + 6 doubling rounds, ~105 us, against the 12 + 1 additions and 6 doublings of one lane
//            each (~310 us) of the scan + tree this replaces.
static __constant__ QuadProgram c_quad_prog[3] = {qp_plain(), qp_weighted(false), qp_weighted(true)};

// One round per iteration: quad q of the wave runs job(q) — the SAME code for every round, a few KB that stay in the
// instruction cache.  `dbl` must be uniform over the wave.
template <class JobFn> __device__ __forceinline__ void quad_exec(TAcc* buf, int nrounds, JobFn fn) {
  const int quad = threadIdx.x >> 2, sub = threadIdx.x & 3;
  for (int r = 0; r < nrounds; r++) {
    bool dbl = false;
    const QuadJob jb = fn(r, quad, dbl);
    TAcc res;
    if (dbl) res = xyzz28_dbl_quad(buf[jb.a]);
    else res = xyzz28_add_quad_mem(buf, jb.a, jb.b);
    if (jb.on && sub == 0) buf[jb.dst] = res;
    __syncthreads();
  }
}

// ------------------------------------------------------------------ table-backed MSM, barrier-free single-wave groups
// 32-copy tables: every radix-256 window of a scalar has its own pre-shifted copy 2^(8c) P of the base, so ALL window
// sums carry weight 1 and a bucket may collect additions from any window: one (scalar, base) pair costs 32 mixed
// additions, the 128 bucket magnitudes of the signed digits are the only reduction.  (Radix 16 with 64 copies — the
// previous layout — paid 60 additions per pair.)
// A single-wave work-group owns WPW consecutive windows of one task and all 128 buckets, two per lane.  With a hit
// rate of 1/128 per (window, point) pair a lane cannot scan for its digits; the wave sorts instead: every lane recodes
// one scalar of a 64-point slab, a histogram over the 128 magnitudes (LDS atomics) gives the list offsets, a second
// pass scatters 16-bit (point, window, sign) entries into per-bucket lists in LDS.  The buckets are then ranked by
// size and lane l takes rank l and rank 127 - l (longest with shortest): the lanes' totals differ by a few additions
// (random pairs: mean 64, slowest lane ~83).  Every lane walks its two lists in ONE loop — gather the pre-shifted
// point from the table in HBM/L2, add; at the boundary it parks the first accumulator in the raw-set buffer.
// A round takes as many slabs as fit the list (8704 entries = 256 points x 32 windows); later rounds of a longer task
// reload the parked accumulators (they are kept per bucket).  Waves never wait for each other and the CU scheduler
// backfills a finished wave's slot.  Output: two raw sets per wave (buckets 1..64, 65..128; k_reduce_sets turns them
// into two partial sums, k_finalize_ranges adds the partials of a request).
constexpr int TBW_WINDOWS = 32;       // radix-256 windows of a scalar < 2^255
constexpr int TBW_CAP = 8704;         // list entries per round
constexpr int TBW_ROUND_PTS = 1024;   // a list entry holds a 10-bit point index beside a 5-bit window and the sign ...
constexpr int TBW_ROUND_PTS_PERWIN = 4096;   // ... or, with one window per scalar half (PERWIN), a 14-bit one: a round takes TBW_CAP / 2 points
static_assert(2 * TBW_ROUND_PTS_PERWIN <= TBW_CAP, "two entries per point of a PERWIN round");
constexpr int TBW_LDS = TBW_CAP * 2 + 3 * 128 * 4 + 128;
constexpr int TBW_LDS_CACHE = TBW_LDS + TBW_ROUND_PTS * 4;   // k_msm_tblw<2, false> keeps a round's recoded digits (below)
static_assert(TBW_CAP >= 64 * TBW_WINDOWS, "one slab always fits");

__device__ __forceinline__ const TAff* tbl_point(const TblTask& t, uint32_t g, uint32_t copy) {
  const uint32_t n0 = t.seg[0].n;
  const TblSeg& sg = (g < n0) ? t.seg[0] : t.seg[1];
  const uint32_t i = (g < n0) ? g : g - n0;
  return sg.base + (size_t)copy * sg.copy_stride + (sg.idx ? sg.idx[i] : i);
}

// Signed radix-256 digits of windows [w0, w0 + WPW) of scalar i, biased by 128 and packed four per word.  The scalar
// is split by the endomorphism first (glv.hpp): k = +-(+-|t| + q z^2), windows 0..15 are the digits of |t| (table copies
// 2^(8c) P), windows 16..31 those of q (table copies -phi(2^(8c) P) = z^2 2^(8c) P); sg_lo / sg_hi are the signs the
// two halves carry.
template <int WPW, bool PERWIN> __device__ __forceinline__ void tbw_digits(const TblTask& task, uint32_t i, int w0, uint32_t (&dg)[(WPW + 3) / 4],
                                                                           uint32_t& sg_lo, uint32_t& sg_hi) {
  if (PERWIN && task.digits) {   // recoded once per task by k_to_table_endo
    const uint32_t* dgs = task.digits + 9 * (size_t)i;
    const uint32_t bt = dgs[w0 >> 2], bq = dgs[4 + (w0 >> 2)], sg = dgs[8];
    dg[0] = ((bt >> (8 * (w0 & 3))) & 255u) | (((bq >> (8 * (w0 & 3))) & 255u) << 8);
    sg_lo = sg & 1u;
    sg_hi = sg >> 1;
    return;
  }
  Fr k = task.scalars[i];
  if (!(task.flags & MSM_SCALARS_CANONICAL)) k = fe_from_mont(k);
  uint32_t t[4], q[4], bytes[8], nk, nt;
  glv_split(k.v, t, q, nk, nt);
  glv_biased_bytes(t, bytes);
  glv_biased_bytes(q, bytes + 4);
  sg_lo = nk ^ nt;
  sg_hi = nk;
  if (PERWIN) {   // w0 = the window: its digit of |t| (entry j = 0) and of q (j = 1)
    uint32_t bt = 0, bq = 0;
    CPX_UNROLL for (int w = 0; w < 4; w++) {
      if (w == (w0 >> 2)) {
        bt = bytes[w];
        bq = bytes[4 + w];
      }
    }
    dg[0] = ((bt >> (8 * (w0 & 3))) & 255u) | (((bq >> (8 * (w0 & 3))) & 255u) << 8);
    return;
  }
  CPX_UNROLL for (int w = 0; w < 8; w++) {
    CPX_UNROLL for (int x = 0; x < (WPW + 3) / 4; x++)
      if (w == (w0 >> 2) + x) dg[x] = bytes[w];
  }
  if (WPW < 4) dg[0] >>= 8 * (w0 & 3);
}
// bucket |d| - 1 of window j's digit, or >= 128 for a zero digit
__device__ __forceinline__ uint32_t tbw_bucket(uint32_t word, int j, uint32_t& neg) {
  const int d = (int)((word >> (8 * (j & 3))) & 255u) - 128;
  neg = d < 0 ? 1u : 0u;
  return (uint32_t)((d < 0 ? -d : d) - 1);   // digit 0 -> 0xffffffff
}

// PERWIN (k_msm_tblw<2, true>): the same wave over bases WITHOUT shifted copies (the verifier's per-proof points, used once):
// the "table" has two copies, P and -phi(P) (k_to_table_endo), wave w of a task takes digit w of |t| and of q, and the
// 16 window results carry the weights 2^(8w) (Horner tail, k_msm_tail).
// PAIR (with PERWIN; the prover's R = <a, vec_R>, S = <a, vec_S>): tasks 2 t and 2 t + 1 have the SAME scalars, hence the same digits
// and the same sorted bucket lists — one wave serves both, walking its lists once per task: the fixed cost of a wave (digit load,
// histogram, ranking, scatter: as much as a dozen additions) is paid once for 16 instead of 8 additions per lane.  Four raw sets
// per wave: [task 2 t lower | upper | task 2 t + 1 lower | upper].
template <int WPW, bool PERWIN, bool PAIR = false> __device__ __forceinline__ void msm_tblw_body(const TblTask* __restrict__ tasks, uint32_t* __restrict__ raw,
                                                                                               uint32_t* __restrict__ raw_slot, int slices, const uint32_t bid) {
  static_assert(!PERWIN || WPW == 2, "one window of each half per wave");
  static_assert(!PAIR || PERWIN, "pairs only for the one-off MSMs");
  constexpr int NSETS = PAIR ? 4 : 2;
  constexpr int WV = PERWIN ? 16 : TBW_WINDOWS / WPW;   // waves per task
  // list entry: (point index inside the round << ESH) | (window j << 1) | sign.  PERWIN has j < 2, so 14 bits of point index fit: the
  // verifier's 1112 points per proof are ONE round instead of 1024 + 88 (the short second round cost a tenth of the kernel: its lists of
  // 0 - 6 entries wait for the longest, and the histogram, ranking and parking are paid again)
  constexpr int ESH = PERWIN ? 2 : 6;
  constexpr uint32_t EJM = PERWIN ? 1u : 31u;
  constexpr uint32_t ROUND_PTS = PERWIN ? (uint32_t)TBW_ROUND_PTS_PERWIN : (uint32_t)TBW_ROUND_PTS;
  constexpr int NW = (WPW + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint16_t* list = reinterpret_cast<uint16_t*>(smem);                 // [TBW_CAP]
  uint32_t* cnt = reinterpret_cast<uint32_t*>(smem + TBW_CAP * 2);    // [128] bucket sizes of the round
  uint32_t* cur = cnt + 128;                                          // [128] scatter cursors
  uint32_t* start = cur + 128;                                        // [128] list offsets
  uint8_t* order = reinterpret_cast<uint8_t*>(start + 128);           // [128] buckets by size
  // the latency form (two windows per wave, a lone proof): the split + recoding of a scalar (~700 instructions) is done in the
  // histogram pass and kept for the scatter pass (one word per point); the other forms recompute it (2 x per 4-32 windows: < 1 %
  // of their additions, and 4 KB more LDS per wave would cost them the second wave per SIMD)
  constexpr bool CACHE = WPW == 2 && !PERWIN;
  uint32_t* dcache = reinterpret_cast<uint32_t*>(smem + TBW_LDS);      // [TBW_ROUND_PTS], CACHE only
  // a few tasks in flight (latency matters): `slices` waves share a task's points (each its own 64-aligned range and raw sets), so
  // that a lane's chain of additions is a quarter as long; slices = 1 otherwise
  const uint32_t tidx = (PAIR ? 2u : 1u) * (bid / (WV * slices));
  const TblTask task = tasks[tidx];
  TblTask task2 = task;   // the second task of a pair: other bases, other partial-sum slots
  if (PAIR) {
    task2.seg[0].base = tasks[tidx + 1].seg[0].base;
    task2.pad = tasks[tidx + 1].pad;
  }
  const int wv = (bid / slices) % WV, w0 = PERWIN ? wv : wv * WPW, slice = bid % slices;
  const int lane = threadIdx.x;
  const uint32_t nall = task.seg[0].n + task.seg[1].n, per = ((nall + slices - 1) / slices + 63) & ~63u;
  const uint32_t first = min(nall, (uint32_t)slice * per), ntot = min(nall, first + per);
  uint32_t* raw0 = raw + (size_t)bid * NSETS * RAW_SET_WORDS;
  if (lane < NSETS) raw_slot[bid * NSETS + lane] = (lane < 2 ? task.pad : task2.pad) + 2 * (wv * slices + slice) + (lane & 1);   // pad = first partial slot of the task

  uint32_t next = first;
  bool later = false;   // a later round: the accumulators are parked in raw0 / raw1
  do {
    cnt[lane] = 0;
    cnt[64 + lane] = 0;
    __syncthreads();
    // histogram over as many 64-point slabs as fit the list
    uint32_t total = 0, end = next;
    while (end < ntot && end - next + 64 <= ROUND_PTS) {
      const uint32_t i = end + lane;
      uint32_t dg[NW], sg_lo, sg_hi;
      int mine = 0;
      if (i < ntot) {
        tbw_digits<WPW, PERWIN>(task, i, w0, dg, sg_lo, sg_hi);
        if (CACHE) dcache[i - next] = (dg[0] & 0xffffu) | (sg_lo << 16) | (sg_hi << 17);
        CPX_UNROLL for (int j = 0; j < WPW; j++) {
          uint32_t neg;
          mine += tbw_bucket(dg[j >> 2], j, neg) < 128u ? 1 : 0;
        }
      }
      int ct = mine;
      CPX_UNROLL for (int m = 32; m >= 1; m >>= 1) ct += __shfl_xor(ct, m, 64);
      if (total && total + (uint32_t)ct > (uint32_t)TBW_CAP) break;
      if (i < ntot) {
        CPX_UNROLL for (int j = 0; j < WPW; j++) {
          uint32_t neg;
          const uint32_t b = tbw_bucket(dg[j >> 2], j, neg);
          if (b < 128u) atomicAdd(&cnt[b], 1u);
        }
      }
      total += (uint32_t)ct;
      end = min(end + 64, ntot);
    }
    __syncthreads();
    // list offsets: exclusive scan of the 128 bucket sizes over the wave (lane l scans buckets 2l, 2l + 1)
    {
      const uint32_t a0 = cnt[2 * lane], a1 = cnt[2 * lane + 1];
      uint32_t incl = a0 + a1;
      CPX_UNROLL for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += y;
      }
      const uint32_t o = incl - (a0 + a1);
      start[2 * lane] = o;
      start[2 * lane + 1] = o + a0;
      cur[2 * lane] = o;
      cur[2 * lane + 1] = o + a0;
      // which two buckets this lane adds up: the 128 buckets ranked by size, lane l takes rank l and rank 127 - l
      // (longest with shortest), so the lanes' totals differ by a few additions instead of +-30 %
      uint32_t r0 = 0, r1 = 0;
      for (int q = 0; q < 128; q++) {
        const uint32_t c = cnt[q];
        r0 += (c < a0 || (c == a0 && q < 2 * lane)) ? 1u : 0u;
        r1 += (c < a1 || (c == a1 && q < 2 * lane + 1)) ? 1u : 0u;
      }
      order[r0] = (uint8_t)(2 * lane);
      order[r1] = (uint8_t)(2 * lane + 1);
    }
    __syncthreads();
    const uint32_t b0 = order[lane];
    const uint32_t c0 = cnt[b0], c1 = cnt[order[127 - lane]], s0_ = start[b0];
    // PAIR: both tasks have the same lists, so a lane that walked its own bucket pair twice carried twice its imbalance (252 points: the
    // longest pair 10.2 entries against a mean of 7.8).  The second task's walk takes the bucket pair of ANOTHER lane instead: the lanes are
    // ranked by the size of their pair and lane of rank r takes the pair of rank 63 - r (the buckets are parked by bucket index, whoever adds
    // them up): the two walks of a lane sum to about twice the mean
    int partner = lane;
    if (PAIR) {
      const uint32_t mine = c0 + c1;
      uint32_t rk = 0;
      for (int q = 0; q < 64; q++) {
        const uint32_t o = (uint32_t)__shfl((int)mine, q, 64);
        rk += (o < mine || (o == mine && q < lane)) ? 1u : 0u;
      }
      // lane of rank 63 - rk: every lane publishes its rank, the partner is found by a second sweep
      for (int q = 0; q < 64; q++) {
        const uint32_t o = (uint32_t)__shfl((int)rk, q, 64);
        if (o == 63u - rk) partner = q;
      }
    }
    for (uint32_t s0 = next; s0 < end; s0 += 64) {
      const uint32_t i = s0 + lane;
      if (i < end) {
        uint32_t dg[NW], sg_lo, sg_hi;
        if (CACHE) {
          const uint32_t v = dcache[i - next];
          dg[0] = v & 0xffffu;
          sg_lo = (v >> 16) & 1u;
          sg_hi = (v >> 17) & 1u;
        } else {
          tbw_digits<WPW, PERWIN>(task, i, w0, dg, sg_lo, sg_hi);
        }
        CPX_UNROLL for (int j = 0; j < WPW; j++) {
          uint32_t neg;
          const uint32_t b = tbw_bucket(dg[j >> 2], j, neg);
          neg ^= (PERWIN ? j == 0 : w0 + j < 16) ? sg_lo : sg_hi;
          if (b < 128u) list[atomicAdd(&cur[b], 1u)] = (uint16_t)(((i - next) << ESH) | ((uint32_t)j << 1) | neg);
        }
      }
    }
    __syncthreads();
    if constexpr (!PAIR) {
      // every lane walks its two lists in one loop; at the boundary it parks the first accumulator (by bucket index)
      const uint32_t b1 = order[127 - lane], s1_ = start[b1];
      uint32_t* const rb0 = raw0 + (b0 >> 6) * RAW_SET_WORDS;   // bucket b lives in set b / 64 at position b % 64
      uint32_t* const rb1 = raw0 + (b1 >> 6) * RAW_SET_WORDS;
      TAcc xacc = later ? raw_load(rb0, b0 & 63) : TAcc::identity();
      const uint32_t cboth = c0 + c1;
      for (uint32_t k = 0; k < cboth; k++) {
        if (k == c0) {   // first bucket done: park it, go on with the second
          raw_store(rb0, b0 & 63, xacc);
          xacc = later ? raw_load(rb1, b1 & 63) : TAcc::identity();
        }
        const uint32_t e = list[k < c0 ? s0_ + k : s1_ + (k - c0)];
        const TAff q = *tbl_point(task, next + (e >> ESH), (PERWIN ? 0u : (uint32_t)w0) + ((e >> 1) & EJM));
        xacc = t_acc_add_mixed_inl(xacc, t_cneg_lazy(q, (e & 1u) != 0));   // products inlined: no argument moves, 231 VGPRs, no scratch
      }
      if (c1 == 0) {
        raw_store(rb0, b0 & 63, xacc);
        xacc = later ? raw_load(rb1, b1 & 63) : TAcc::identity();
      }
      raw_store(rb1, b1 & 63, xacc);
    } else {
      // ONE loop over the lane's four bucket lists — its own two for the first task, the two of its partner lane for the second: a wave runs a
      // loop until its slowest lane is through, so only inside one loop do a lane's long and short lists add up (two loops, one per task,
      // each waited for the longest pair of buckets of the wave, whoever walked it; the same loop in the one-task kernels measured 0.4 % slower
      // than the two-list loop above).  Segment sg: bucket, list, raw set and task; a segment that ends parks its accumulator by bucket index
      // and loads (or zeroes) the next one.
      constexpr int NSEG = PAIR ? 4 : 2;
      auto seg_bucket = [&](int sg) { return (uint32_t)order[(sg & 1) ? 127 - ((PAIR && sg >= 2) ? partner : lane) : ((PAIR && sg >= 2) ? partner : lane)]; };
      uint32_t nwalk = c0 + c1;   // entries of all the lane's lists
      if (PAIR) nwalk += cnt[seg_bucket(2)] + cnt[seg_bucket(3)];
      const ptrdiff_t base2 = PAIR ? task2.seg[0].base - task.seg[0].base : 0;   // (a pair's tasks differ in their bases only)
      int sg = 0;
      uint32_t bk = b0, cs = c0, st = s0_, kin = 0;
      uint32_t* rb = raw0 + (bk >> 6) * RAW_SET_WORDS;   // bucket b lives in set b / 64 (+ 2 for the second task of a pair) at position b % 64
      TAcc xacc = later ? raw_load(rb, bk & 63) : TAcc::identity();
      auto next_segment = [&]() {
        raw_store(rb, bk & 63, xacc);
        sg++;
        bk = seg_bucket(sg);
        cs = cnt[bk];
        st = start[bk];
        kin = 0;
        rb = raw0 + (2 * (sg >> 1) + (bk >> 6)) * RAW_SET_WORDS;
        xacc = later ? raw_load(rb, bk & 63) : TAcc::identity();
      };
#pragma unroll 1
      for (uint32_t k = 0; k < nwalk; k++) {
        while (kin == cs) next_segment();   // (sg stays below NSEG: entries remain)
        const uint32_t e = list[st + kin];
        const TAff* src = tbl_point(task, next + (e >> ESH), (PERWIN ? 0u : (uint32_t)w0) + ((e >> 1) & EJM));
        if (PAIR && sg >= 2) src += base2;
        const TAff q = *src;
        xacc = t_acc_add_mixed_inl(xacc, t_cneg_lazy(q, (e & 1u) != 0));   // products inlined: no argument moves, no scratch
        kin++;
      }
      while (sg + 1 < NSEG) next_segment();   // the lists that were empty or remain: every bucket of the round is written
      raw_store(rb, bk & 63, xacc);
    }
    __syncthreads();
    next = end;
    later = true;
  } while (next < ntot);
}
// ------------------------------------------------------------------ fixed-base MSM over multiples tables
constexpr int FIX_CHUNK = 256;

// signed radix-2^c digits of a canonical scalar (recode.hpp); writes the NW digits of windows [w0, w0+NW).  Digits in
// [-2^(c-1), 2^(c-1) - 1]: |d| indexes the table, -2^15 still fits int16_t; 19-bit digits need 32 bits
template <int CB> struct FixDigit { typedef int16_t type; };
template <> struct FixDigit<19> { typedef int32_t type; };
template <int CB, int NW> __device__ __forceinline__ void fix_digits(const uint32_t* k, int w0, typename FixDigit<CB>::type* out, int stride) {
  fix_window_digits<CB, NW>(k, w0, out, stride);   // recode.hpp (host + device: checked on the CPU by tests/test_host_emul.py)
}
// One wave = FIX_WPW windows x (64 / FIX_WPW) point slices of one task.  16 windows per wave (a whole radix-2^16
// task) keeps the fixed 6-step tree reduction small against the per-lane additions (n/4 per lane); 8 windows
// per wave double the number of waves when the launch would not fill the GPU otherwise.
template <int CB, int FIX_WPW> __device__ __forceinline__ void msm_fix_body(const FixTask* __restrict__ tasks, const TFix* __restrict__ tab, int nc,
                                                                          uint32_t* __restrict__ raw, uint32_t* __restrict__ raw_slot, const uint32_t bid) {
  constexpr int W = (256 + CB - 1) / CB, WG = W / FIX_WPW;    // waves per task
  constexpr int LPW = 64 / FIX_WPW;                // lanes (point slices) per window (CB = 19: 7 windows x 9 slices, lane 63 idles)
  constexpr uint32_t M = 1u << (CB - 1);
  static_assert(W % FIX_WPW == 0, "whole waves per task");
  typedef typename FixDigit<CB>::type DT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  DT* digits = reinterpret_cast<DT*>(smem);   // [FIX_WPW][FIX_CHUNK]
  const FixTask task = tasks[bid / WG];
  const int wg = bid % WG;
  const int lane = threadIdx.x;
  const bool working = lane < FIX_WPW * LPW;
  const int wl = working ? lane / LPW : 0, sl = lane % LPW;
  const uint32_t w = (uint32_t)(wg * FIX_WPW + wl);
  const TFix* wtab = tab + (size_t)w * M * nc;
  TAcc xacc = TAcc::identity();

  for (uint32_t base = 0; base < task.n; base += FIX_CHUNK) {
    const int m = min((uint32_t)FIX_CHUNK, task.n - base);
    __syncthreads();
    for (int t = lane; t < m; t += 64) {
      Fr k = task.scalars[base + t];
      if (!(task.flags & MSM_SCALARS_CANONICAL)) k = fe_from_mont(k);
      fix_digits<CB, FIX_WPW>(k.v, wg * FIX_WPW, digits + t, FIX_CHUNK);
    }
    __syncthreads();
    const DT* row = digits + wl * FIX_CHUNK;
    // every lane walks its slice i = sl, sl+LPW, ...: same trip count for all lanes; the gather of the next point is in
    // flight while the current one is added (products inlined)
    auto fetch = [&](int i, int& d) {
      d = (i < m && working) ? row[i] : 0;
      const uint32_t g = base + i;
      return d ? wtab[(size_t)((d < 0 ? -d : d) - 1) * nc + task.off + (task.idx ? task.idx[g] : g)].a : TAff::identity();
    };
    int dn;
    TAff pn = fetch(sl, dn);
    for (int i = sl; i < m; i += LPW) {
      const int d = dn;
      const TAff p = pn;
      pn = fetch(i + LPW, dn);
      if (d) xacc = t_acc_add_mixed_inl(xacc, t_cneg_lazy(p, d < 0));
    }
  }
  // all entries carry weight 1: the wave's 64 accumulators form one plain raw set (k_reduce_sets adds them up)
  raw_store(raw + (size_t)bid * RAW_SET_WORDS, lane, xacc);
  if (lane == 0) raw_slot[bid] = task.out_first + wg;
}
}  // namespace cpx
