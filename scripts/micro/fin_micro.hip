// Stage timing of the wave-per-request finalisation (k_finalize_ranges_wave of kernels.hip, copied with time stamps): where the
// ~300 us of a lone proof's finalisation go.  Build like lat_micro.hip (includes the product's kernels.hip for its device functions).
#include "../../curdleproofs_amd/csrc/kernels.hip"
#include <cstdio>
using namespace cpx;
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int RPB> __global__ __launch_bounds__(64 * RPB) void k_fin_timed(const TJac* __restrict__ part, const uint32_t* __restrict__ first, const uint32_t* __restrict__ count, int n,
                                                   Aff* __restrict__ out_aff, uint8_t* __restrict__ out_comp, uint64_t* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  TF* buf = reinterpret_cast<TF*>(smem);
  TJac* red = reinterpret_cast<TJac*>(smem + 2 * 64 * RPB * sizeof(TF));
  const int tid = threadIdx.x, lane = tid & 63, g = blockIdx.x * RPB + (tid >> 6);
  uint64_t s0 = wall_clock64();
  TJac p = TJac::identity();
  uint32_t cmax = 1;
  for (int q = 0; q < RPB; q++)
    if (blockIdx.x * RPB + q < n) cmax = max(cmax, count[blockIdx.x * RPB + q]);
  if (g < n) {
    const TJac* src = part + first[g];
    const uint32_t c = count[g];
    if ((uint32_t)lane < c) p = src[lane];
    for (uint32_t j = lane + 64; j < c; j += 64) p = t_add(p, src[j]);
  }
  red[tid] = p;
  __syncthreads();
  uint64_t s1 = wall_clock64();
  int top = 32;
  while (top >= 1 && (uint32_t)top >= cmax) top >>= 1;
  for (int sft = top; sft >= 1; sft >>= 1) {
    const bool has = lane < sft;
    TJac other;
    if (has) other = red[tid + sft];
    __syncthreads();
    if (has) {
      p = t_add(p, other);
      red[tid] = p;
    }
    __syncthreads();
  }
  uint64_t s2 = wall_clock64();
  const bool owner = lane == 0 && g < n;
  const bool inf = p.is_identity();
  const TF zinv = t_block_batch_inverse(owner ? p.z : t_one(), buf);
  uint64_t s3 = wall_clock64();
  if (tid == 0) { stamps[0] = s1 - s0; stamps[1] = s2 - s1; stamps[2] = s3 - s2; }
  if (!owner) return;
  const Aff a = inf ? Aff::identity() : t_to_std(t_to_affine(p, zinv));
  uint64_t s4 = wall_clock64();
  out_aff[g] = a;
  aff_compress_store(a, out_comp + (size_t)g * 48);
  uint64_t s5 = wall_clock64();
  if (tid == 0) { stamps[3] = s4 - s3; stamps[4] = s5 - s4; }
}

int main() {
  const int n = 4, cnt = 16;
  std::vector<TJac> h(n * cnt);
  for (size_t i = 0; i < h.size(); i++) {
    int32_t* w = reinterpret_cast<int32_t*>(&h[i]);
    for (size_t j = 0; j < sizeof(TJac) / 4; j++) w[j] = (int32_t)((0x9e3779b1u * (uint32_t)(i * 131 + j + 7)) & 0x0fffffff);
  }
  std::vector<uint32_t> first(n), count(n, cnt);
  for (int i = 0; i < n; i++) first[i] = i * cnt;
  TJac* dp; uint32_t *df, *dc; Aff* da; uint8_t* dcomp; uint64_t* dt;
  HIPCHECK(hipMalloc(&dp, h.size() * sizeof(TJac))); HIPCHECK(hipMalloc(&df, 4 * n)); HIPCHECK(hipMalloc(&dc, 4 * n));
  HIPCHECK(hipMalloc(&da, n * sizeof(Aff))); HIPCHECK(hipMalloc(&dcomp, 48 * n)); HIPCHECK(hipMalloc(&dt, 64));
  HIPCHECK(hipMemcpy(dp, h.data(), h.size() * sizeof(TJac), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(df, first.data(), 4 * n, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(dc, count.data(), 4 * n, hipMemcpyHostToDevice));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rpb : {4, 1}) {
    const size_t lds = 2 * 64 * rpb * sizeof(TF) + 64 * rpb * sizeof(TJac);
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(a, 0);
      if (rpb == 4) hipLaunchKernelGGL(k_fin_timed<4>, dim3(1), dim3(256), lds, 0, dp, df, dc, n, da, dcomp, dt);
      else hipLaunchKernelGGL(k_fin_timed<1>, dim3(4), dim3(64), lds, 0, dp, df, dc, n, da, dcomp, dt);
      hipEventRecord(b, 0);
      HIPCHECK(hipDeviceSynchronize());
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    uint64_t t[5];
    HIPCHECK(hipMemcpy(t, dt, sizeof t, hipMemcpyDeviceToHost));
    printf("requests per block %d: load %.1f us | tree(%d partials) %.1f us | batch inverse %.1f us | to_affine+std %.1f us | compress+store %.1f us | kernel (events) %.1f us\n",
           rpb, t[0] / 100.0, cnt, t[1] / 100.0, t[2] / 100.0, t[3] / 100.0, t[4] / 100.0, ms * 1e3);
  }
  return 0;
}
