// gfx950 VALU micro-benchmarks for the big-integer inner loop (standalone; not part of libcpx).
//   hipcc --offload-arch=gfx950 -O3 valu_micro.hip -o valu_micro && ./valu_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// 1 chain of dependent v_mad_u64_u32
__global__ void k_mad_dep(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x;
  uint32_t va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) {
    asm volatile(REP64("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n") : "+v"(acc) : "v"(va), "v"(vb) : "s20", "s21");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// 2 independent chains
__global__ void k_mad_2(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t x = threadIdx.x, y = threadIdx.x + 7;
  uint32_t va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) {
    asm volatile(REP8(REP8("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\nv_mad_u64_u32 %1, s[22:23], %2, %3, %1\n")) : "+v"(x), "+v"(y) : "v"(va), "v"(vb) : "s20", "s21", "s22", "s23");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x ^ y;
}
// 4 independent chains
__global__ void k_mad_4(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t x = threadIdx.x, y = threadIdx.x + 7, z = 3, w = 9;
  uint32_t va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) {
    asm volatile(REP8(REP8("v_mad_u64_u32 %0, s[20:21], %4, %5, %0\nv_mad_u64_u32 %1, s[22:23], %4, %5, %1\nv_mad_u64_u32 %2, s[24:25], %4, %5, %2\nv_mad_u64_u32 %3, s[26:27], %4, %5, %3\n"))
                 : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(va), "v"(vb) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x ^ y ^ z ^ w;
}
// dependent simple VALU (v_add_u32)
__global__ void k_add_dep(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint32_t acc = threadIdx.x, va = a;
  for (int i = 0; i < iters; i++) {
    asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(acc) : "v"(va));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// 4 independent simple VALU
__global__ void k_add_4(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint32_t x = threadIdx.x, y = 1, z = 2, w = 3, va = a;
  for (int i = 0; i < iters; i++) {
    asm volatile(REP8(REP8("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4\n")) : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(va));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x ^ y ^ z ^ w;
}
// the MAC pair as used in fe_mul: mad ; s_nop 1 ; addc   (dependent chain)
__global__ void k_mac_nop(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x;
  uint32_t c2 = 0, va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) {
    asm volatile(REP64("v_mad_u64_u32 %0, vcc, %2, %3, %0\ns_nop 1\nv_addc_co_u32 %1, vcc, 0, %1, vcc\n") : "+v"(acc), "+v"(c2) : "v"(va), "v"(vb) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + c2;
}
// two interleaved MAC chains, carries through two SGPR pairs, no nops:  madX madY addcX' addcY' (carry of previous step)
__global__ void k_mac_2(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t x = threadIdx.x, y = threadIdx.x + 5;
  uint32_t cx = 0, cy = 0, va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) {
    asm volatile(REP8(REP8("v_mad_u64_u32 %0, s[20:21], %4, %5, %0\nv_mad_u64_u32 %2, s[22:23], %4, %5, %2\ns_nop 0\nv_addc_co_u32 %1, vcc, 0, %1, s[20:21]\nv_addc_co_u32 %3, vcc, 0, %3, s[22:23]\n"))
                 : "+v"(x), "+v"(cx), "+v"(y), "+v"(cy) : "v"(va), "v"(vb) : "vcc", "s20", "s21", "s22", "s23");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + cx + cy;
}
// v_mul_lo_u32 + v_mul_hi_u32 dependent
__global__ void k_mullohi(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint32_t x = threadIdx.x + 3, y = 7, va = a | 1;
  for (int i = 0; i < iters; i++) {
    asm volatile(REP8(REP8("v_mul_lo_u32 %0, %0, %2\nv_mul_hi_u32 %1, %1, %2\n")) : "+v"(x), "+v"(y) : "v"(va));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x ^ y;
}

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t, int);
struct K { const char* name; kern_t k; int ops_per_rep; };

int main() {
  uint64_t* d;
  HIPCHECK(hipMalloc(&d, 256 * 32 * 256 * 8 * 8));
  K ks[] = {{"mad_dep (1 chain)", k_mad_dep, 64}, {"mad_2 chains", k_mad_2, 128}, {"mad_4 chains", k_mad_4, 256},
            {"add_dep", k_add_dep, 64}, {"add_4 chains", k_add_4, 256}, {"mac: mad,s_nop1,addc", k_mac_nop, 64},
            {"mac x2 interleaved (per MAC)", k_mac_2, 128}, {"mul_lo+mul_hi (per op)", k_mullohi, 128}};
  int iters = 2000;
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  int clk = 0;
  HIPCHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
  printf("clock attr %d kHz\n", clk);
  for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD
    int blocks = 256 * 4 * wps / 4;          // 256-thread blocks (4 waves): wps blocks per CU
    printf("-- %d wave(s) per SIMD (%d blocks of 256)\n", wps, blocks);
    for (auto& kk : ks) {
      hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, d, 12345u, 678u, 10);
      HIPCHECK(hipDeviceSynchronize());
      HIPCHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, d, 12345u, 678u, iters);
      HIPCHECK(hipEventRecord(e1));
      HIPCHECK(hipDeviceSynchronize());
      float ms;
      HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
      double ops_per_wave = (double)iters * kk.ops_per_rep;
      double ns_per_op = ms * 1e6 / ops_per_wave;                 // per wave-instruction, as seen by ONE wave
      double simd_ns_per_op = ns_per_op / wps;                    // SIMD time per wave-instruction
      printf("%-32s %8.3f ms   %6.2f ns/op/wave   %6.2f ns SIMD-time/op  (~%.1f cyc @2.4GHz)\n", kk.name, ms, ns_per_op, simd_ns_per_op, simd_ns_per_op * 2.4);
    }
  }
  return 0;
}
