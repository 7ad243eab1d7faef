// gfx950 latency micro-benchmark of the pieces of the single-wave chains (k_reduce_sets_wave / k_finalize_ranges_wave): how long
// one wave needs for a dependent chain of table-form point additions, for the division-step inversion, for the conversions and the
// compression.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I curdleproofs_amd/csrc scripts/micro/lat_micro.hip -o scripts/micro/lat_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "g1.hpp"
#include "g1_28.hpp"
using namespace cpx;
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __noinline__ TF inv_dev(const TF& a) { return t_inv(a); }

__global__ void k(const Aff* in, uint64_t* t, Aff* out, uint8_t* comp, int reps) {
  const int lane = threadIdx.x;
  TAff a = t_from_std(in[lane]);
  TJac p = t_add_mixed(TJac::identity(), a), q = t_dbl(p);
  uint64_t t0 = wall_clock64();
  for (int i = 0; i < reps; i++) p = t_add(p, q);           // dependent Jacobian additions
  uint64_t t1 = wall_clock64();
  for (int i = 0; i < reps; i++) p = t_add_mixed(p, a);     // dependent mixed additions
  uint64_t t2 = wall_clock64();
  for (int i = 0; i < reps; i++) p = t_dbl(p);
  uint64_t t3 = wall_clock64();
  TF z = p.z;
  for (int i = 0; i < 4; i++) z = inv_dev(t_mul(z, p.x));   // 4 dependent inversions (+ one product each)
  uint64_t t4 = wall_clock64();
  TF m = z;
  for (int i = 0; i < 64; i++) m = t_mul(m, p.y);           // 64 dependent products
  uint64_t t5 = wall_clock64();
  const TAff r = t_to_affine(p, m);
  const Aff s = t_to_std(r);
  uint64_t t6 = wall_clock64();
  aff_compress(s, comp + 48 * lane);
  uint64_t t7 = wall_clock64();
  out[lane] = s;
  if (lane == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; t[5] = t6 - t5; t[6] = t7 - t6; }
}

int main() {
  Aff h[64];
  for (int i = 0; i < 64; i++) {   // any field values do for timing (formulas are branch-free on values)
    for (int j = 0; j < 12; j++) { h[i].x.v[j] = 0x1234567u * (i + 1) + j; h[i].y.v[j] = 0x7654321u * (i + 3) + 5 * j; }
    h[i].x.v[11] &= 0x0fffffff; h[i].y.v[11] &= 0x0fffffff;
  }
  Aff *din, *dout; uint64_t* dt; uint8_t* dc;
  HIPCHECK(hipMalloc(&din, sizeof h)); HIPCHECK(hipMalloc(&dout, sizeof h)); HIPCHECK(hipMalloc(&dt, 64)); HIPCHECK(hipMalloc(&dc, 48 * 64));
  HIPCHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
  const int reps = 32;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dt, dout, dc, reps);
    HIPCHECK(hipDeviceSynchronize());
  }
  uint64_t t[7];
  HIPCHECK(hipMemcpy(t, dt, sizeof t, hipMemcpyDeviceToHost));
  const double us = 1.0 / 100.0;   // wall_clock64: 100 MHz
  printf("t_add        %.2f us each\n", t[0] * us / reps);
  printf("t_add_mixed  %.2f us each\n", t[1] * us / reps);
  printf("t_dbl        %.2f us each\n", t[2] * us / reps);
  printf("t_inv (+mul) %.2f us each\n", t[3] * us / 4);
  printf("t_mul        %.2f us each\n", t[4] * us / 64);
  printf("to_affine + to_std  %.2f us\n", t[5] * us);
  printf("aff_compress        %.2f us\n", t[6] * us);
  return 0;
}
