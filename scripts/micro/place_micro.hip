// Where do single-wave work-groups land?  Prints (XCC, SE, CU, SIMD) per work-group for a plain stream and for a stream created with
// a CU mask (upper half of the bits), and what happens to a second kernel launched beside a long-running first one.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/place_micro.hip -o scripts/micro/place_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(64) void k(uint32_t* out, long spin) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint64_t t0 = wall_clock64();
  while ((long)(wall_clock64() - t0) < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
static void show(const char* name, const std::vector<uint32_t>& h, int n) {
  printf("%s\n", name);
  for (int i = 0; i < n; i++) {
    const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    printf("  wg %2d: xcc %u se %u sh %u cu %2u simd %u wave %u\n", i, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
  }
}
int main() {
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount, n = 40;
  printf("CUs %d\n", ncu);
  uint32_t *d1, *d2;
  HIPCHECK(hipMalloc(&d1, 8 * 256)); HIPCHECK(hipMalloc(&d2, 8 * 256));
  hipStream_t plain, plain2, masked;
  HIPCHECK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  HIPCHECK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
  std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
  for (int cu = ncu / 2; cu < ncu; cu++) mask[cu / 32] |= 1u << (cu % 32);
  HIPCHECK(hipExtStreamCreateWithCUMask(&masked, (uint32_t)mask.size(), mask.data()));
  std::vector<uint32_t> h(512);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, plain, d1, 1000L);
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpy(h.data(), d1, 8 * n, hipMemcpyDeviceToHost));
  show("plain stream, 40 work-groups", h, n);
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, masked, d1, 1000L);
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpy(h.data(), d1, 8 * n, hipMemcpyDeviceToHost));
  show("masked stream (upper half of the CU bits), 40 work-groups", h, n);
  // a long kernel on one plain stream, then a second one beside it
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, plain, d1, 200000L);   // 2 ms
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, plain2, d2, 1000L);
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpy(h.data(), d2, 8 * n, hipMemcpyDeviceToHost));
  show("second kernel beside a running 40-group kernel (both plain)", h, n);
  return 0;
}
