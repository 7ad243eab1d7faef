// Batched-affine accumulation against the extended-Jacobian chain of k_msm_fix — measured, not sized.
//
//   A  what the MSM kernels do: a lane adds its table entries one after the other into an XYZZ accumulator (madd-2008-s with the shared
//      reduction of Y3: 3542 multiply-adds = 9.04 products per addition, g1_28.hpp t_acc_add_mixed_inl), N entries per set.
//   B  a tree of AFFINE additions (lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1: 2M + 1S + the
//      inverse): the denominators of a tree level — PAIRS per lane, 64 lanes — go through ONE inversion per wave (per-lane Montgomery
//      trick + the wave-wide product scans and the division-step inversion of block_inverse.hpp, exactly what the finalisation kernels use).
//      At the register budget of two waves per SIMD (256 VGPRs) a lane holds four affine points (112 registers): PAIRS = 2, a set of
//      4 entries is 2 + 1 additions in two levels, i.e. two wave-wide inversions per set.
// Both variants gather the same pseudo-random entries of a table of multiples of the generator (table form, 128-byte lines, resident in L2
// like the hot part of the real table: the gathers are hidden either way, profiles/r04_gather_micro.txt), at 2 waves per SIMD, and write
// one result per set; B's results are checked against A's (same group elements).
// What bounds B is not the field products but the inversion: a wave executes its instruction stream for all 64 lanes, so "sharing" one
// inversion over the wave saves nothing in issue slots — its ~78 us of single-lane latency (5.5 mixed additions) are paid per level and wave.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I curdleproofs_amd/csrc scripts/micro/affine_micro.hip -o scripts/micro/affine_micro
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "g1.hpp"
#include "g1_28.hpp"
#include "block_inverse.hpp"

using namespace cpx;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int NT = 8192;   // table entries: (i + 1) G
struct alignas(128) Line {
  TAff a;
};

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// entry for (lane, set, i): distinct entries inside a set (no P + P, no P - P in the affine formulas: the real kernel would need the checks too)
__device__ __forceinline__ uint32_t entry_of(uint32_t g, uint32_t set, uint32_t i, uint32_t n) { return (mix(g * 7919u + set) * n + i * 37u + (mix(g + set * 31u) % 37u)) % NT; }

__global__ __launch_bounds__(64) void k_mk(Aff g, Line* tab) {
  __shared__ TF buf[2 * 64];
  const int i = blockIdx.x * 64 + threadIdx.x;
  const TAff ga = t_from_std(g);
  TJac p = TJac::identity();
  for (int b = 13; b >= 0; b--) {
    p = t_dbl(p);
    if (((i + 1) >> b) & 1) p = t_add_mixed(p, ga);
  }
  const TF zinv = t_block_batch_inverse(p.z, buf);
  tab[i].a = t_to_affine(p, zinv);
}

// A: XYZZ chain, N entries per set
template <int N> __global__ __launch_bounds__(64, 2) void k_chain(const Line* __restrict__ tab, int sets, TAff* __restrict__ out) {
  __shared__ TF buf[2 * 64];
  const uint32_t g = blockIdx.x * 64 + threadIdx.x;
  for (int s = 0; s < sets; s++) {
    TAcc acc = TAcc::identity();
    TAff nxt = tab[entry_of(g, s, 0, N)].a;
    for (int i = 0; i < N; i++) {
      const TAff p = nxt;
      if (i + 1 < N) nxt = tab[entry_of(g, s, i + 1, N)].a;
      acc = t_acc_add_mixed_inl(acc, p);
    }
    if (out) {   // (checking run: normalise so that the two variants can be compared)
      const TJac j = t_acc_to_jac(acc);
      const TF zinv = t_block_batch_inverse(j.z, buf);
      out[(size_t)s * gridDim.x * 64 + g] = t_to_affine(j, zinv);
    } else {
      reinterpret_cast<volatile int32_t*>(buf)[threadIdx.x] = acc.x.v[0] ^ acc.zz.v[3];   // keep the result alive
    }
  }
}

// affine addition given 1 / (x2 - x1)
__device__ __forceinline__ TAff aff_add_with_inv(const TAff& p, const TAff& q, const TF& dinv) {
  const TF lam = t_mul(f28_sub(q.y, p.y), dinv);
  const TF x3 = f28_sub(f28_sub(f28_sqr(lam), p.x), q.x);
  const TF y3 = f28_sub(t_mul(lam, f28_sub(p.x, x3)), p.y);
  return TAff{x3, y3};
}

// B: tree of affine additions over 4 entries per set (two pairs per lane), one inversion per wave and level
__global__ __launch_bounds__(64, 2) void k_tree(const Line* __restrict__ tab, int sets, TAff* __restrict__ out) {
  __shared__ TF buf[2 * 64];
  const uint32_t g = blockIdx.x * 64 + threadIdx.x;
  for (int s = 0; s < sets; s++) {
    const TAff p0 = tab[entry_of(g, s, 0, 4)].a, p1 = tab[entry_of(g, s, 1, 4)].a, p2 = tab[entry_of(g, s, 2, 4)].a, p3 = tab[entry_of(g, s, 3, 4)].a;
    // level 1: (p0 + p1), (p2 + p3): the lane's two denominators through one inversion per wave
    const TF d0 = f28_sub(p1.x, p0.x), d1 = f28_sub(p3.x, p2.x);
    const TF inv01 = t_block_batch_inverse(t_mul(d0, d1), buf);
    const TAff a = aff_add_with_inv(p0, p1, t_mul(inv01, d1)), b = aff_add_with_inv(p2, p3, t_mul(inv01, d0));
    // level 2: a + b
    const TF dinv = t_block_batch_inverse(f28_sub(b.x, a.x), buf);
    const TAff r = aff_add_with_inv(a, b, dinv);
    if (out) out[(size_t)s * gridDim.x * 64 + g] = r;
    else reinterpret_cast<volatile int32_t*>(buf)[threadIdx.x] = r.x.v[0] ^ r.y.v[3];
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int sets = argc > 1 ? atoi(argv[1]) : 16;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 4 * 2;   // 2 waves per SIMD
  Aff gpt;
  const uint32_t gx[12] = {0xfd530c16, 0x5cb38790, 0x9976fff5, 0x7817fc67, 0x143ba1c1, 0x154f95c7, 0xf3d0e747, 0xf0ae6acd, 0x21dbf440, 0xedce6ecc, 0x9e0bfb75, 0x12017741};
  const uint32_t gy[12] = {0x0ce72271, 0xbaac93d5, 0x7918fd8e, 0x8c22631a, 0x570725ce, 0xdd595f13, 0x50405194, 0x51ac5829, 0xad0059c0, 0x0e1c8c3f, 0x5008a26a, 0x0bbc3efc};
  memcpy(gpt.x.v, gx, 48);
  memcpy(gpt.y.v, gy, 48);
  Line* tab;
  CK(hipMalloc(&tab, sizeof(Line) * NT));
  hipLaunchKernelGGL(k_mk, dim3(NT / 64), dim3(64), 0, 0, gpt, tab);
  CK(hipDeviceSynchronize());
  // correctness: the same 4 entries per set through both variants
  {
    const int cb = 8, cs = 4;
    TAff *oa, *ob;
    CK(hipMalloc(&oa, sizeof(TAff) * cb * 64 * cs));
    CK(hipMalloc(&ob, sizeof(TAff) * cb * 64 * cs));
    hipLaunchKernelGGL(k_chain<4>, dim3(cb), dim3(64), 0, 0, tab, cs, oa);
    hipLaunchKernelGGL(k_tree, dim3(cb), dim3(64), 0, 0, tab, cs, ob);
    CK(hipDeviceSynchronize());
    std::vector<TAff> ha(cb * 64 * cs), hb(cb * 64 * cs);
    CK(hipMemcpy(ha.data(), oa, sizeof(TAff) * ha.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), ob, sizeof(TAff) * hb.size(), hipMemcpyDeviceToHost));
    // (both are lazily reduced table-form values: compare after the conversion to the standard form on the host side of g1_28.hpp)
    int bad = 0;
    for (size_t i = 0; i < ha.size(); i++) {
      const Aff x = t_to_std(ha[i]), y = t_to_std(hb[i]);
      if (memcmp(&x, &y, sizeof x)) bad++;
    }
    printf("check: %zu sums of 4 entries, affine tree against XYZZ chain: %d mismatches\n", ha.size(), bad);
    if (bad) return 1;
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  auto time_it = [&](auto launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(a));
      launch();
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    return best;
  };
  printf("%s, %d single-wave groups (2 per SIMD), %d sets per lane\n", prop.name, blocks, sets);
  const double lanes = (double)blocks * 64;
  const float t16 = time_it([&] { hipLaunchKernelGGL(k_chain<16>, dim3(blocks), dim3(64), 0, 0, tab, sets, (TAff*)nullptr); });
  const float t4 = time_it([&] { hipLaunchKernelGGL(k_chain<4>, dim3(blocks), dim3(64), 0, 0, tab, sets * 4, (TAff*)nullptr); });
  const float tb = time_it([&] { hipLaunchKernelGGL(k_tree, dim3(blocks), dim3(64), 0, 0, tab, sets * 4, (TAff*)nullptr); });
  printf("A  XYZZ chain, 16 entries per set (k_msm_fix's loop)   %8.3f ms  %7.2f G additions/s (16 per set)\n", t16, lanes * sets * 16 / t16 / 1e6);
  printf("A' XYZZ chain,  4 entries per set                      %8.3f ms  %7.2f G entries/s\n", t4, lanes * sets * 4 * 4 / t4 / 1e6);
  printf("B  affine tree, 4 entries per set, 2 inversions / wave %8.3f ms  %7.2f G entries/s   B / A' = %.2f x the time\n", tb, lanes * sets * 4 * 4 / tb / 1e6, tb / t4);
  printf("   per set and wave: B %.1f us, A' %.1f us — the two wave-wide inversions alone are ~2 x (12 scan products + the division-step inversion)\n",
         tb * 1e3 / (sets * 4) / 2, t4 * 1e3 / (sets * 4) / 2);
  return 0;
}
