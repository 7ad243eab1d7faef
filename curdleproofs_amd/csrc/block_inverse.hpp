// In-work-group batch inversion on the device (shared by kernels.hip and late.hip) — product code.
// One inversion per work-group by batched division steps (modinv30.hpp): ~0.05 ms of single-wave latency instead of
// the ~0.25 ms of the binary Euclidean algorithm or the ~0.55 ms of Fermat's a^(p-2) (571 dependent products);
// callers amortise it over a whole work-group.
#pragma once
#include <hip/hip_runtime.h>
#include "g1.hpp"
#include "g1_28.hpp"

namespace cpx {

static __device__ __noinline__ Fp fp_inv_device(const Fp& a) { return fe_inv_divsteps(a); }

// Every thread passes its z (any value; zero is treated as 1 and returns an unspecified value the caller must
// ignore).  `buf` holds 2 * blockDim.x Fp values.  Inclusive Hillis-Steele product scans from both ends, one
// inversion by wave 0.
static __device__ __forceinline__ Fp block_batch_inverse(Fp z, Fp* buf) {
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

// The same on the table form (fp28.hpp): 28-bit-limb lazy values.
static __device__ __noinline__ TF t_inv_device(const TF& a) { return t_inv(a); }
static __device__ __forceinline__ TF t_block_batch_inverse(TF z, TF* buf) {
  const int tid = threadIdx.x, n = blockDim.x;
  if (t_is_zero_raw(z)) z = t_one();
  TF* pre = buf;
  TF* suf = buf + n;
  TF p = z, s = z;
  pre[tid] = p;
  suf[tid] = s;
  __syncthreads();
  for (int off = 1; off < n; off <<= 1) {
    TF pl, sr;
    const bool hp = tid >= off, hs = tid + off < n;
    if (hp) pl = pre[tid - off];
    if (hs) sr = suf[tid + off];
    __syncthreads();
    if (hp) p = t_mul(p, pl);
    if (hs) s = t_mul(s, sr);
    pre[tid] = p;
    suf[tid] = s;
    __syncthreads();
  }
  const TF left = (tid > 0) ? pre[tid - 1] : t_one();
  const TF right = (tid + 1 < n) ? suf[tid + 1] : t_one();
  TF tinv = t_one();
  if (tid < 64) tinv = t_inv_device(pre[n - 1]);
  __syncthreads();
  if (tid == 0) pre[0] = tinv;
  __syncthreads();
  tinv = pre[0];
  return t_mul(t_mul(tinv, left), right);
}

}  // namespace cpx
