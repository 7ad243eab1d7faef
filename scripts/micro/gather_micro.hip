// How fast can a wave gather random 128-byte table entries (the access pattern of k_msm_fix / k_late_fix over the 17.5 GB fixed-base
// table: one padded 128-byte line per mixed addition), and does it matter HOW the 112 bytes of an entry are fetched?
//   A  every lane fetches its own entry with 7 x global_load_dwordx4 (what the kernels do): 64 different lines — and 64 address
//      translations — per load instruction, 448 per wave-gather;
//   B  eight lanes fetch ONE entry together (lane j of the group loads bytes [16 j, 16 j + 16)): 8 full lines per load instruction,
//      8 instructions per wave-gather, the pieces find their owner through LDS (8 ds_write_b128 + 7 ds_read_b128 per lane).
// Each variant runs alone (memory only) and with a dependent multiply-add chain of `work` instructions per gather beside it (a mixed
// addition is ~5300 instructions), at 2 waves per SIMD like the MSM kernels, over a table of `gb` GB and over a cache-resident one.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_micro.hip -o scripts/micro/gather_micro
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint64_t row_of(uint32_t seed, uint64_t nrows) {
  const uint64_t r = ((uint64_t)mix(seed) << 32) | mix(seed ^ 0x9e3779b9u);
  return r % nrows;
}
// `work` dependent multiply-adds (the VALU work between two gathers)
__device__ __forceinline__ uint64_t busy(uint64_t acc, uint32_t a, int work) {
#pragma unroll 1
  for (int i = 0; i < work; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; j++) acc = (uint64_t)(uint32_t)acc * a + acc;
  }
  return acc;
}

template <bool COOP> __global__ __launch_bounds__(64, 2) void k_gather(const uint4* __restrict__ tab, uint64_t nrows, int iters, int work, uint64_t* __restrict__ out) {
  __shared__ uint4 stage[64 * 8];
  __shared__ uint64_t rows[64];
  const int lane = threadIdx.x;
  const uint32_t base = (blockIdx.x * 64 + lane) * 2654435761u;
  uint64_t acc = base | 1;
  uint4 cur[7];
  auto issue = [&](int it, uint4 (&dst)[8]) {
    const uint64_t r = row_of(base + it, nrows);
    if (!COOP) {
      const uint4* p = tab + r * 8;
#pragma unroll
      for (int k = 0; k < 7; k++) dst[k] = p[k];
    } else {
      rows[lane] = r;
      __builtin_amdgcn_wave_barrier();
      const int grp = lane & ~7, sub = lane & 7;
#pragma unroll
      for (int k = 0; k < 8; k++) dst[k] = tab[rows[grp + k] * 8 + sub];   // lane `sub` of the group: bytes [16 sub, 16 sub + 16) of entry grp + k
    }
  };
  auto land = [&](uint4 (&src)[8]) {
    if (!COOP) {
#pragma unroll
      for (int k = 0; k < 7; k++) cur[k] = src[k];
    } else {
      const int grp = lane & ~7, sub = lane & 7;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 8; k++) stage[(grp + k) * 8 + sub] = src[k];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 7; k++) cur[k] = stage[lane * 8 + k];
    }
  };
  uint4 nxt[8];
  issue(0, nxt);
  for (int it = 0; it < iters; it++) {
    land(nxt);
    if (it + 1 < iters) issue(it + 1, nxt);   // the next entry is in flight while this one is "added"
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) x ^= cur[k].x ^ cur[k].y ^ cur[k].z ^ cur[k].w;
    acc = busy(acc + x, x | 1, work);
  }
  out[blockIdx.x * 64 + lane] = acc;
}

__global__ void k_fill(uint4* tab, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t v = mix((uint32_t)i);
    tab[i] = make_uint4(v, v ^ 1, v ^ 2, v ^ 3);
  }
}

int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 16.0;
  const int iters = argc > 2 ? atoi(argv[2]) : 256;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 4 * 2;   // 2 waves per SIMD
  uint64_t* out;
  CK(hipMalloc(&out, (size_t)blocks * 64 * 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  printf("%s, %d CUs, %d single-wave groups (2 per SIMD), %d gathers per lane\n", prop.name, cus, blocks, iters);
  for (double size_gb : {gb, 0.03125}) {
    const uint64_t nrows = (uint64_t)(size_gb * (1ull << 30) / 128);
    uint4* tab;
    CK(hipMalloc(&tab, nrows * 128));
    k_fill<<<4096, 256>>>(tab, nrows * 8);
    CK(hipDeviceSynchronize());
    printf("-- table %.3f GB (%llu entries of 128 B)\n", size_gb, (unsigned long long)nrows);
    for (int work : {0, 1024, 2560, 5120}) {
      for (int coop = 0; coop < 2; coop++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
          CK(hipEventRecord(a));
          if (coop) k_gather<true><<<blocks, 64>>>(tab, nrows, iters, work, out);
          else k_gather<false><<<blocks, 64>>>(tab, nrows, iters, work, out);
          CK(hipEventRecord(b));
          CK(hipEventSynchronize(b));
          float ms;
          CK(hipEventElapsedTime(&ms, a, b));
          if (ms < best) best = ms;
        }
        const double gathers = (double)blocks * 64 * iters;
        printf("work %5d mads/gather  %s  %8.3f ms  %7.2f G gathers/s  %7.1f GB/s  %6.2f us per gather and wave\n", work,
               coop ? "B 8 lanes per entry + LDS" : "A lane per entry (7 x b128)", best, gathers / best / 1e6, gathers * 128 / best / 1e6, best * 1e3 / iters);
      }
    }
    CK(hipFree(tab));
  }
  return 0;
}
