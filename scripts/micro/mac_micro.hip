// gfx950 micro-benchmarks of candidate multiply-accumulate patterns for the Montgomery inner loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// A: current MAC (mad ; s_nop 1 ; addc via vcc)
__global__ void kA(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x; uint32_t c2 = 0, va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) asm volatile(REP64("v_mad_u64_u32 %0, vcc, %2, %3, %0\ns_nop 1\nv_addc_co_u32 %1, vcc, 0, %1, vcc\n") : "+v"(acc), "+v"(c2) : "v"(va), "v"(vb) : "vcc");
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + c2;
}
// B: same without the nop (hazard-unsafe; timing only)
__global__ void kB(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x; uint32_t c2 = 0, va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) asm volatile(REP64("v_mad_u64_u32 %0, vcc, %2, %3, %0\nv_addc_co_u32 %1, vcc, 0, %1, vcc\n") : "+v"(acc), "+v"(c2) : "v"(va), "v"(vb) : "vcc");
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + c2;
}
// C: software-pipelined, one dependent accumulator chain, carries through 3 rotating SGPR pairs, no nops:
//    mad1 mad2 [mad3 addc1] [mad4 addc2] ...   (addc two mads after its producer)
#define C3 "v_mad_u64_u32 %0, s[20:21], %2, %3, %0\nv_addc_co_u32 %1, vcc, 0, %1, s[24:25]\n" \
           "v_mad_u64_u32 %0, s[22:23], %2, %3, %0\nv_addc_co_u32 %1, vcc, 0, %1, s[20:21]\n" \
           "v_mad_u64_u32 %0, s[24:25], %2, %3, %0\nv_addc_co_u32 %1, vcc, 0, %1, s[22:23]\n"
__global__ void kC(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x; uint32_t c2 = 0, va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) asm volatile(REP8(C3 C3 C3 C3 C3 C3 C3 C3) : "+v"(acc), "+v"(c2) : "v"(va), "v"(vb) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25");
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + c2;
}
// D: like C but the addc writes its (unused) carry-out to a scratch SGPR pair instead of vcc
#define D3 "v_mad_u64_u32 %0, s[20:21], %2, %3, %0\nv_addc_co_u32 %1, s[26:27], 0, %1, s[24:25]\n" \
           "v_mad_u64_u32 %0, s[22:23], %2, %3, %0\nv_addc_co_u32 %1, s[26:27], 0, %1, s[20:21]\n" \
           "v_mad_u64_u32 %0, s[24:25], %2, %3, %0\nv_addc_co_u32 %1, s[26:27], 0, %1, s[22:23]\n"
__global__ void kD(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x; uint32_t c2 = 0, va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) asm volatile(REP8(D3 D3 D3 D3 D3 D3 D3 D3) : "+v"(acc), "+v"(c2) : "v"(va), "v"(vb) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + c2;
}
// E: carry-free column of 14 mads + limb extraction (28-bit limbs): and, mul_lo, and, mad, lshr_b64
#define E14 REP8("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n") "v_mad_u64_u32 %0, s[20:21], %1, %2, %0\nv_mad_u64_u32 %0, s[20:21], %1, %2, %0\nv_mad_u64_u32 %0, s[20:21], %1, %2, %0\nv_mad_u64_u32 %0, s[20:21], %1, %2, %0\nv_mad_u64_u32 %0, s[20:21], %1, %2, %0\nv_mad_u64_u32 %0, s[20:21], %1, %2, %0\n"
__global__ void kE(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x; uint32_t va = a + threadIdx.x, vb = b, m = 0;
  for (int i = 0; i < iters; i++)
    asm volatile(REP8(E14 "v_and_b32 %3, 0xfffffff, %1\nv_mul_lo_u32 %3, %3, %2\nv_and_b32 %3, 0xfffffff, %3\nv_mad_u64_u32 %0, s[20:21], %3, %2, %0\nv_lshrrev_b64 %0, 28, %0\n")
                 : "+v"(acc), "+v"(va) : "v"(vb), "v"(m) : "s20", "s21");
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// F: s_nop cost: mad ; s_nop 0
__global__ void kF(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x; uint32_t va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) asm volatile(REP64("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\ns_nop 0\n") : "+v"(acc) : "v"(va), "v"(vb) : "s20", "s21");
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// G: mad ; s_nop 1
__global__ void kG(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x; uint32_t va = a + threadIdx.x, vb = b;
  for (int i = 0; i < iters; i++) asm volatile(REP64("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\ns_nop 1\n") : "+v"(acc) : "v"(va), "v"(vb) : "s20", "s21");
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// H: v_addc alone (vcc chain, unsafe) ; I: v_lshrrev_b64 alone ; J: mad with SGPR multiplicand
__global__ void kH(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint32_t c2 = threadIdx.x;
  for (int i = 0; i < iters; i++) asm volatile(REP64("v_addc_co_u32 %0, vcc, 0, %0, vcc\n") : "+v"(c2) : : "vcc");
  out[blockIdx.x * blockDim.x + threadIdx.x] = c2;
}
__global__ void kI(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x * 77777ull + a;
  for (int i = 0; i < iters; i++) asm volatile(REP64("v_lshrrev_b64 %0, 1, %0\n") : "+v"(acc));
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void kJ(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t acc = threadIdx.x; uint32_t va = a + threadIdx.x;
  for (int i = 0; i < iters; i++) asm volatile(REP64("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n") : "+v"(acc) : "v"(va), "s"(b) : "s20", "s21");
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t, int);
struct K { const char* name; kern_t k; double units; };
int main() {
  uint64_t* d;
  HIPCHECK(hipMalloc(&d, 256 * 32 * 256 * 8));
  K ks[] = {{"A mac: mad,s_nop1,addc(vcc)      /MAC", kA, 64}, {"B mac: mad,addc no nop (unsafe)  /MAC", kB, 64},
            {"C pipelined 3 sgpr pairs, no nop /MAC", kC, 192}, {"D same, addc carry-out to sgpr   /MAC", kD, 192},
            {"E carry-free 28-bit column       /mad", kE, 8 * 15}, {"F mad,s_nop0                     /pair", kF, 64},
            {"G mad,s_nop1                     /pair", kG, 64}, {"H addc alone                     /op", kH, 64},
            {"I lshrrev_b64                    /op", kI, 64}, {"J mad with sgpr operand          /op", kJ, 64}};
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  int iters = 1000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    int blocks = 256 * wps;
    printf("-- %d wave(s) per SIMD\n", wps);
    for (auto& kk : ks) {
      hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, d, 12345u, 678u, 10);
      HIPCHECK(hipDeviceSynchronize());
      HIPCHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, d, 12345u, 678u, iters);
      HIPCHECK(hipEventRecord(e1));
      HIPCHECK(hipDeviceSynchronize());
      float ms;
      HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
      double ns = ms * 1e6 / ((double)iters * kk.units) / wps;
      printf("%-42s %7.3f ms  SIMD-time %.2f ns = %.1f cyc @2.4GHz\n", kk.name, ms, ns, ns * 2.4);
    }
  }
  return 0;
}
