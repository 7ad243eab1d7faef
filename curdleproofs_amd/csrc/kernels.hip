// gfx950 (MI355X / CDNA4) kernels for the Curdleproofs G1 hot path.
//
// Replaces, behind include/cpx.h, the arithmetic the reference reaches through
//   util::msm                       /root/reference/src/util.rs:19-22            -> k_msm_accum + k_msm_tail
//   IPA / SameMSM basis folds       inner_product_argument.rs:174-179,
//                                   same_multiscalar_argument.rs:126-131           -> k_smul (add operand set)
//   G' rescale / shuffle scaling    grand_product_argument.rs:90-102, util.rs:94-95 -> k_smul (no add operand)
//   into_affine / normalize_batch + compressed serialisation                       -> k_finalize
//
// Integer big-number work on the 32-bit VALU (v_mad_u64_u32 chains); no MFMA — there is no dense
// contraction here.  Design notes:
//  * wave = 64 lanes; MSM work-group = 512 threads = 64 windows x 8 signed-digit buckets, so every
//    lane owns exactly one Pippenger bucket in registers (36 VGPRs) and walks the window's digit row
//    in LDS; the points of the current chunk are staged once in LDS and shared by all 64 windows.
//  * the long doubling chain that combines windows is split out of the bucket kernel into
//    k_msm_tail, one *lane* per partial Horner evaluation, so that lanes stay busy across the many
//    independent MSMs of a proof batch instead of one lane per work-group idling 511 others.
//  * per-element scalar multiplications share the scalar across a whole wave in the fold case, so
//    the NAF-driven add branch is wave-uniform; the Jacobian results are normalised with one Fp
//    inversion per work-group (LDS product scans) instead of the reference's one inversion per point.
#include <hip/hip_runtime.h>
#include "g1.cuh"
#include "recode.cuh"
#include "kernels.h"

namespace cpx {

// ------------------------------------------------------------------ inversion on device
// Fermat a^(p-2) with the exponent words held in SGPRs/literals.  One call costs ~570 Fp products,
// so callers amortise it over a whole work-group (batch inversion below).
__device__ __noinline__ Fp fp_inv_device(const Fp& a) {
  Fp r = Fp::one();
  bool started = false;
  for (int l = 11; l >= 0; l--) {
    uint32_t w = FpCfg::P[l];
    if (l == 0) w -= 2;
    for (int b = 31; b >= 0; b--) {
      if (started) r = fe_sqr(r);
      if ((w >> b) & 1u) {
        r = started ? fe_mul(r, a) : a;
        started = true;
      }
    }
  }
  return r;
}

// In-work-group batch inversion.  Every thread passes its z (any value; zero is treated as 1 and
// returns an unspecified value the caller must ignore).  `buf` holds 2 * blockDim.x Fp values.
// Inclusive Hillis-Steele product scans from both ends, one Fermat inversion by wave 0.
__device__ __forceinline__ Fp block_batch_inverse(Fp z, Fp* buf) {
  const int tid = threadIdx.x, n = blockDim.x;
  if (z.is_zero()) z = Fp::one();
  Fp* pre = buf;        // inclusive prefix products
  Fp* suf = buf + n;    // inclusive suffix products
  Fp p = z, s = z;
  pre[tid] = p;
  suf[tid] = s;
  __syncthreads();
  for (int off = 1; off < n; off <<= 1) {
    Fp pl, sr;
    const bool hp = tid >= off, hs = tid + off < n;
    if (hp) pl = pre[tid - off];
    if (hs) sr = suf[tid + off];
    __syncthreads();
    if (hp) p = fe_mul(p, pl);
    if (hs) s = fe_mul(s, sr);
    pre[tid] = p;
    suf[tid] = s;
    __syncthreads();
  }
  // z_i^-1 = total^-1 * pre[i-1] * suf[i+1]
  const Fp left = (tid > 0) ? pre[tid - 1] : Fp::one();
  const Fp right = (tid + 1 < n) ? suf[tid + 1] : Fp::one();
  Fp tinv = Fp::one();
  if (tid < 64) tinv = fp_inv_device(pre[n - 1]);   // one wave; its lanes compute it redundantly
  __syncthreads();                                  // all reads of pre/suf are done
  if (tid == 0) pre[0] = tinv;
  __syncthreads();
  tinv = pre[0];
  return fe_mul(fe_mul(tinv, left), right);
}

// ------------------------------------------------------------------ MSM: bucket accumulation
// One work-group per task.  Thread t owns bucket (window w = t / 8, magnitude b = t % 8 + 1).
// LDS: digit rows [64][MSM_CHUNK] int8, staged points [MSM_CHUNK] Aff; afterwards the same LDS is
// reused for the 512 bucket sums (Jacobian) that wave 0 folds into 64 window sums.
constexpr int MSM_THREADS = 512;
constexpr int MSM_CHUNK = 256;
constexpr int MSM_LDS_BYTES = 512 * (int)sizeof(Jac);   // 73,728 B >= 64*256 + 256*96 = 40,960 B

#ifndef CPX_MSM_WAVES
#define CPX_MSM_WAVES 2   // waves per SIMD the register allocator must leave room for (2 = one 512-thread group per CU)
#endif
__global__ __launch_bounds__(MSM_THREADS, CPX_MSM_WAVES) void k_msm_accum(const MsmTask* __restrict__ tasks, Jac* __restrict__ wsum) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int8_t* digits = reinterpret_cast<int8_t*>(smem);                       // [64][MSM_CHUNK]
  Aff* pts = reinterpret_cast<Aff*>(smem + 64 * MSM_CHUNK);               // [MSM_CHUNK]
  Jac* buckets = reinterpret_cast<Jac*>(smem);                            // [512] (aliases the above later)

  const MsmTask task = tasks[blockIdx.x];
  const int tid = threadIdx.x;
  const int w = tid >> 3, mag = (tid & 7) + 1;
  Jac acc = Jac::identity();

  for (uint32_t base = 0; base < task.n; base += MSM_CHUNK) {
    const int m = min((uint32_t)MSM_CHUNK, task.n - base);
    __syncthreads();   // previous chunk fully consumed
    if (tid < m) {
      Fr k = task.scalars[base + tid];
      if (!(task.flags & MSM_SCALARS_CANONICAL)) k = fe_from_mont(k);
      recode_signed16(k.v, digits + tid, MSM_CHUNK);
    } else if (tid >= 256 && tid - 256 < m) {
      const uint32_t j = base + tid - 256;
      pts[tid - 256] = task.bases[task.idx ? task.idx[j] : j];
    }
    __syncthreads();
    const int8_t* row = digits + w * MSM_CHUNK;
    int i = 0;
    for (;;) {
      // cheap divergent scan to this lane's next matching point
      int d = 0;
      while (i < m) {
        d = row[i];
        if (d == mag || d == -mag) break;
        i++;
      }
      const bool have = i < m;
      if (!__any(have)) break;
      if (have) {
        Aff q = pts[i];
        acc = jac_add_mixed(acc, aff_cneg(q, d < 0));
        i++;
      }
    }
  }
  __syncthreads();
  buckets[tid] = acc;
  __syncthreads();
  // Window sum  sum_{b=1..8} b * B_b  =  sum_b (suffix sum of B from b): a Hillis-Steele suffix scan over the
  // window's 8 lanes (3 steps) followed by a tree sum (3 steps) — 6 point additions with every wave of the
  // work-group busy, instead of 16 on one wave with seven waves idle.
  const int b = tid & 7;
  for (int sft = 1; sft < 8; sft <<= 1) {
    const bool has = b + sft < 8;
    Jac other;
    if (has) other = buckets[tid + sft];
    __syncthreads();
    if (has) {
      acc = jac_add(acc, other);
      buckets[tid] = acc;
    }
    __syncthreads();
  }
  for (int sft = 4; sft >= 1; sft >>= 1) {
    const bool has = b < sft;
    Jac other;
    if (has) other = buckets[tid + sft];
    __syncthreads();
    if (has) {
      acc = jac_add(acc, other);
      buckets[tid] = acc;
    }
    __syncthreads();
  }
  if (b == 0) wsum[(size_t)blockIdx.x * 64 + w] = acc;
}

// ------------------------------------------------------------------ MSM: Horner tails
// out[t] = sum_{j < group} 2^(shift*j) in[t*group + j], one lane per output.
__global__ __launch_bounds__(64) void k_msm_tail(const Jac* __restrict__ in, Jac* __restrict__ out, int nout, int group, int shift) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nout) return;
  const Jac* src = in + (size_t)t * group;
  Jac acc = src[group - 1];
  for (int j = group - 2; j >= 0; j--) {
    for (int s = 0; s < shift; s++) acc = jac_dbl(acc);
    acc = jac_add(acc, src[j]);
  }
  out[t] = acc;
}

// ------------------------------------------------------------------ normalise + compress
// Thread per point: Jacobian -> affine (one inversion per work-group) and, optionally, the 48-byte
// compressed encoding the transcript and the proof bytes need.
__global__ __launch_bounds__(256) void k_finalize(const Jac* __restrict__ in, int n, Aff* __restrict__ out_aff, const uint32_t* __restrict__ dst_index,
                                                   uint8_t* __restrict__ out_comp) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  Fp* buf = reinterpret_cast<Fp*>(smem);
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  Jac p = Jac::identity();
  if (g < n) p = in[g];
  const bool inf = p.is_identity();
  Fp zinv = block_batch_inverse(p.z, buf);
  if (g >= n) return;
  Aff a = inf ? Aff::identity() : jac_to_affine_with_zinv(p, zinv);
  if (out_aff) out_aff[dst_index ? dst_index[g] : g] = a;
  if (out_comp) aff_compress(a, out_comp + (size_t)g * 48);
}

// affine -> compressed bytes only; `batch` strided input rows of n points each, dense output
__global__ __launch_bounds__(256) void k_compress(const Aff* __restrict__ in, int n, int stride, uint8_t* __restrict__ out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const size_t row = blockIdx.y;
  aff_compress(in[row * stride + g], out + (row * n + g) * 48);
}

// ------------------------------------------------------------------ per-element scalar-mul (+ add)
// Element e of task t:  out[e] = affine( (add ? add[e] : 0) + s * mul[e] ),
//   s = scalars[e * scalar_stride]  (stride 0: one scalar for the whole task = a fold).
// All tasks of a launch have `cnt` elements; consecutive threads take consecutive elements so the
// 96-byte point loads/stores of a wave are contiguous.
constexpr int SMUL_THREADS = 256;

#ifndef CPX_SMUL_WAVES
#define CPX_SMUL_WAVES 2
#endif
__global__ __launch_bounds__(SMUL_THREADS, CPX_SMUL_WAVES) void k_smul(const SmulTask* __restrict__ tasks, int ntasks, int cnt) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  Fp* buf = reinterpret_cast<Fp*>(smem);                                             // 2*256 Fp = 24 KB
  uint32_t* naf = reinterpret_cast<uint32_t*>(smem + 2 * SMUL_THREADS * sizeof(Fp));  // [18][256]
  const int tid = threadIdx.x;
  const long g = (long)blockIdx.x * blockDim.x + tid;
  const long total = (long)ntasks * cnt;
  const bool live = g < total;
  Jac acc = Jac::identity();
  Aff* dst = nullptr;
  if (live) {
    const int t = (int)(g / cnt), e = (int)(g % cnt);
    const SmulTask task = tasks[t];
    dst = task.out + e;
    Fr k = task.scalars[(size_t)e * task.scalar_stride];
    if (!(task.flags & MSM_SCALARS_CANONICAL)) k = fe_from_mont(k);
    uint32_t nz[9], ng[9];
    recode_naf(k.v, nz, ng);
    CPX_UNROLL for (int i = 0; i < 9; i++) {
      naf[i * SMUL_THREADS + tid] = nz[i];
      naf[(9 + i) * SMUL_THREADS + tid] = ng[i];
    }
    const Aff P = task.mul[e];
    for (int i = 256; i >= 0; i--) {
      acc = jac_dbl_fn(acc);
      const uint32_t nzw = naf[(i >> 5) * SMUL_THREADS + tid];
      if ((nzw >> (i & 31)) & 1u) {
        const uint32_t ngw = naf[(9 + (i >> 5)) * SMUL_THREADS + tid];
        acc = jac_add_mixed_fn(acc, aff_cneg(P, (ngw >> (i & 31)) & 1u));
      }
    }
    if (task.add) acc = jac_add_mixed_fn(acc, task.add[e]);
  }
  const bool inf = acc.is_identity();
  Fp zinv = block_batch_inverse(acc.z, buf);
  if (!live) return;
  *dst = inf ? Aff::identity() : jac_to_affine_with_zinv(acc, zinv);
}

// ------------------------------------------------------------------ decompression (+ subgroup check)
// ark-bls12-381 `deserialize_compressed` (Validate::Yes) for tracker / proof ingestion
// (/root/reference/src/whisk.rs:265-277, :318-320; curdleproofs.rs:312-323): y = sqrt(x^3 + 4) by
// a^((p+1)/4), sign chosen by the flag bit, then r * P == O.
__device__ __noinline__ Fp fp_sqrt_candidate(const Fp& a) {
  // (p+1)/4, little-endian 32-bit words
  const uint32_t E[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                          0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};
  Fp r = Fp::one();
  bool started = false;
  for (int l = 11; l >= 0; l--) {
    const uint32_t w = E[l];
    for (int b = 31; b >= 0; b--) {
      if (started) r = fe_sqr(r);
      if ((w >> b) & 1u) {
        r = started ? fe_mul(r, a) : a;
        started = true;
      }
    }
  }
  return r;
}

__global__ __launch_bounds__(64) void k_decompress(const uint8_t* __restrict__ in, int n, Aff* __restrict__ out,
                                                   const uint32_t* __restrict__ dst_index, uint8_t* __restrict__ status, int check_subgroup) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const uint8_t* b = in + (size_t)g * 48;
  const bool compressed = b[0] & 0x80, infinity = b[0] & 0x40, sort = b[0] & 0x20;
  Fp xc;
  for (int l = 0; l < 12; l++) {
    uint32_t w = ((uint32_t)b[4 * l] << 24) | ((uint32_t)b[4 * l + 1] << 16) | ((uint32_t)b[4 * l + 2] << 8) | b[4 * l + 3];
    if (l == 0) w &= 0x1fffffffu;
    xc.v[11 - l] = w;
  }
  Aff res = Aff::identity();
  uint8_t st = 0;
  if (!compressed) {
    st = 1;
  } else if (infinity) {
    if (sort || !xc.is_zero()) st = 1;
  } else {
    Fp pm;   // x must be < p
    for (int i = 0; i < 12; i++) pm.v[i] = FpCfg::P[i];
    if (!fe_raw_gt(pm, xc)) {
      st = 1;
    } else {
      const Fp x = fe_to_mont(xc);
      Fp four = Fp::one();
      four = fe_dbl(fe_dbl(four));
      const Fp rhs = fe_add(fe_mul(fe_sqr(x), x), four);
      Fp y = fp_sqrt_candidate(rhs);
      if (!fe_sqr(y).eq(rhs)) {
        st = 1;
      } else {
        const Fp ny = fe_neg(y);
        const bool y_larger = fe_raw_gt(fe_from_mont(y), fe_from_mont(ny));
        res.x = x;
        res.y = (y_larger == sort) ? y : ny;
        if (check_subgroup) {
          // r * P by MSB-first double-and-add over the fixed modulus bits
          Jac acc = Jac::identity();
          for (int l = 7; l >= 0; l--) {
            const uint32_t w = FrCfg::P[l];
            for (int bit = 31; bit >= 0; bit--) {
              acc = jac_dbl(acc);
              if ((w >> bit) & 1u) acc = jac_add_mixed(acc, res);
            }
          }
          if (!acc.is_identity()) st = 2;
        }
      }
    }
  }
  out[dst_index ? dst_index[g] : g] = st ? Aff::identity() : res;
  status[g] = st;
}

// ------------------------------------------------------------------ micro-benchmarks (bench.py --micro)
// Dependent Fp multiply chains: measures the achieved modular-multiply rate of this build.
__global__ __launch_bounds__(256) void k_bench_fpmul(Fp* data, int iters) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  Fp x = data[g], y = data[g + gridDim.x * blockDim.x];
  for (int i = 0; i < iters; i++) {
    x = fe_mul(x, y);
    y = fe_mul(y, x);
  }
  data[g] = fe_add(x, y);
}

// ------------------------------------------------------------------ host-callable launchers
void launch_msm_accum(const MsmTask* d_tasks, int ntasks, Jac* d_wsum, hipStream_t s) {
  if (ntasks <= 0) return;
  hipLaunchKernelGGL(k_msm_accum, dim3(ntasks), dim3(MSM_THREADS), MSM_LDS_BYTES, s, d_tasks, d_wsum);
}
void launch_msm_tail(const Jac* d_in, Jac* d_out, int nout, int group, int shift, hipStream_t s) {
  if (nout <= 0) return;
  hipLaunchKernelGGL(k_msm_tail, dim3((nout + 63) / 64), dim3(64), 0, s, d_in, d_out, nout, group, shift);
}
void launch_finalize(const Jac* d_in, int n, Aff* d_aff, const uint32_t* d_dst_index, uint8_t* d_comp, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_finalize, dim3((n + 255) / 256), dim3(256), 2 * 256 * sizeof(Fp), s, d_in, n, d_aff, d_dst_index, d_comp);
}
void launch_compress(const Aff* d_in, int n, int stride, int batch, uint8_t* d_out, hipStream_t s) {
  if (n <= 0 || batch <= 0) return;
  hipLaunchKernelGGL(k_compress, dim3((n + 255) / 256, batch), dim3(256), 0, s, d_in, n, stride, d_out);
}
void launch_smul(const SmulTask* d_tasks, int ntasks, int cnt, hipStream_t s) {
  const long total = (long)ntasks * cnt;
  if (total <= 0) return;
  const size_t lds = 2 * SMUL_THREADS * sizeof(Fp) + 18 * SMUL_THREADS * sizeof(uint32_t);
  hipLaunchKernelGGL(k_smul, dim3((unsigned)((total + SMUL_THREADS - 1) / SMUL_THREADS)), dim3(SMUL_THREADS), lds, s, d_tasks, ntasks, cnt);
}
void launch_decompress(const uint8_t* d_in, int n, Aff* d_out, const uint32_t* d_dst_index, uint8_t* d_status, int check_subgroup, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_decompress, dim3((n + 63) / 64), dim3(64), 0, s, d_in, n, d_out, d_dst_index, d_status, check_subgroup);
}
void launch_bench_fpmul(Fp* d_data, int blocks, int iters, hipStream_t s) {
  hipLaunchKernelGGL(k_bench_fpmul, dim3(blocks), dim3(256), 0, s, d_data, iters);
}

}  // namespace cpx
