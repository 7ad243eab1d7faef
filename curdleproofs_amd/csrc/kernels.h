// Task descriptors and launchers of the gfx950 kernels (kernels.hip) — shared with the host engine.
#pragma once
#include <hip/hip_runtime.h>
#include "g1.cuh"

namespace cpx {

enum : uint32_t { MSM_SCALARS_CANONICAL = 1u };

// One multi-scalar multiplication  sum_i scalars[i] * bases[idx ? idx[i] : i],  i < n.
// All pointers are device pointers.  Scalars are Fr in Montgomery form unless flagged canonical.
struct MsmTask {
  const Aff* bases;
  const uint32_t* idx;   // optional gather list into `bases`
  const Fr* scalars;
  uint32_t n;
  uint32_t flags;
};

// cnt elements:  out[e] = affine( (add ? add[e] : O) + scalars[e * scalar_stride] * mul[e] )
struct SmulTask {
  const Aff* add;        // nullable
  const Aff* mul;
  Aff* out;
  const Fr* scalars;
  uint32_t scalar_stride;   // 0 = one shared scalar (fold), 1 = per element (rescale)
  uint32_t flags;
};

// ntasks MSMs: bucket accumulation -> d_wsum[ntasks*64] window sums (Jacobian) ...
void launch_msm_accum(const MsmTask* d_tasks, int ntasks, Jac* d_wsum, hipStream_t s);
// ... then Horner tails: out[t] = sum_{j<group} 2^(shift*j) in[t*group+j]; (8,4) then (8,32) gives the MSM results
void launch_msm_tail(const Jac* d_in, Jac* d_out, int nout, int group, int shift, hipStream_t s);
// Jacobian -> affine (+ optional scatter through dst_index) and optional 48-byte compressed form.
void launch_finalize(const Jac* d_in, int n, Aff* d_aff, const uint32_t* d_dst_index, uint8_t* d_comp, hipStream_t s);
// `batch` rows of n points, row r starting at d_in + r*stride; output dense [batch][n][48]
void launch_compress(const Aff* d_in, int n, int stride, int batch, uint8_t* d_out, hipStream_t s);
void launch_smul(const SmulTask* d_tasks, int ntasks, int cnt, hipStream_t s);
// compressed bytes -> affine; status[i] = 0 ok, 1 malformed / not on curve, 2 not in the r-torsion subgroup
void launch_decompress(const uint8_t* d_in, int n, Aff* d_out, const uint32_t* d_dst_index, uint8_t* d_status, int check_subgroup, hipStream_t s);
void launch_bench_fpmul(Fp* d_data, int blocks, int iters, hipStream_t s);

}  // namespace cpx
