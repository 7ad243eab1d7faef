// gfx950 latency of the lane-parallel Keccak-f[1600] of wave_strobe.hpp (one wave per state, bit-interleaved halves) and of a rate block
// absorbed from memory; checks one permutation and a 1000-byte absorb against the host code of strobe.hpp.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I curdleproofs_amd/csrc scripts/micro/keccak_micro.hip -o scripts/micro/keccak_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include "wave_strobe.hpp"
using namespace cpx;
#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k(const uint8_t* msg, size_t len, uint64_t* t, uint64_t* out, int reps) {
  __shared__ uint8_t scratch[64];
  WaveStrobe s;
  s.set_lane(threadIdx.x);
  s.init("bench", 5, scratch);
  uint64_t t0 = wall_clock64();
  for (int i = 0; i < reps; i++) s.keccak();
  uint64_t t1 = wall_clock64();
  s.absorb(msg, len);
  uint64_t t2 = wall_clock64();
  if (s.w >= 0) out[blockIdx.x * 64 + (threadIdx.x & 32) + s.w] = s.a;   // (keeps the work alive)
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; }
}
__global__ __launch_bounds__(64) void k_check(const uint64_t* in27, uint64_t* out27, const uint8_t* msg, size_t len) {
  WaveStrobe s;
  s.load(in27, threadIdx.x);
  s.keccak();
  s.store(out27);
  // a message absorbed from an odd position, then one challenge-style squeeze of the first 32 bytes
  s.load(in27, threadIdx.x);
  s.absorb(msg, len);
  s.xor_byte(s.pos, 0xa7);
  s.store(out27 + 27);
  uint64_t sq[4];
  s.peek4(sq);
  if (threadIdx.x == 63) for (int j = 0; j < 4; j++) out27[54 + j] = sq[j];
  // a run of Merlin operations: a message with its label and length, a scalar, three challenges (positions 17 .. wrap around the rate)
  __shared__ uint8_t scratch[64];
  s.load(in27, threadIdx.x);
  for (int rep = 0; rep < 3; rep++) {
    s.append_message("some_label", 10, msg, 48, scratch);
    Fr x;
    for (int j = 0; j < 8; j++) x.v[j] = 0x01234567u * (j + 1 + rep);
    x.v[7] &= 0x0fffffffu;
    s.append_scalar("a_scalar", 8, fe_to_mont(x), scratch);
    const Fr c = s.challenge_scalar("a_challenge", 11, scratch);
    if (threadIdx.x == 5) for (int j = 0; j < 4; j++) out27[58 + 27 + 4 * rep + j] = (uint64_t)c.v[2 * j] | ((uint64_t)c.v[2 * j + 1] << 32);
  }
  s.store(out27 + 58);
}
// where a challenge's time goes: the steps of WaveStrobe::challenge_attempt timed one by one (wall clock, 10 ns ticks), 252 challenges
__global__ __launch_bounds__(64) void k_sections(const uint64_t* in27, uint64_t* tsum, Fr* out, int n) {
  __shared__ uint8_t scratch[64];
  WaveStrobe t;
  t.load(in27, threadIdx.x);
  uint64_t acc[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    Fr c;
    const uint64_t t0 = wall_clock64();
    t.begin_op(Strobe::FLAG_M | Strobe::FLAG_A, false);
    t.absorb(reinterpret_cast<const uint8_t*>("curdleproofs_vec_a"), 18, 64, 4);
    const uint64_t t1 = wall_clock64();
    t.begin_op(Strobe::FLAG_I | Strobe::FLAG_A | Strobe::FLAG_C, false);
    const uint64_t t2 = wall_clock64();
    uint64_t sq[4];
    t.peek4(sq);
    if (t.w >= 0 && t.w < 8) t.a = 0;
    t.pos = 64;
    sq[3] &= 0x7fffffffffffffffULL;
    for (int j = 0; j < 4; j++) { c.v[2 * j] = (uint32_t)sq[j]; c.v[2 * j + 1] = (uint32_t)(sq[j] >> 32); }
    const uint64_t t3 = wall_clock64();
    t.append_begin("curdleproofs_vec_a", 18, 32, scratch);
    const uint64_t t4 = wall_clock64();
    t.absorb_words8(c.v);
    const uint64_t t5 = wall_clock64();
    if (threadIdx.x == 0) out[i] = fe_to_mont(c);
    const uint64_t t6 = wall_clock64();
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; acc[4] += t5 - t4; acc[5] += t6 - t5;
  }
  t.store(tsum + 8);
  if (threadIdx.x == 0 && blockIdx.x == 0) for (int j = 0; j < 6; j++) tsum[j] = acc[j];
}
// the transcript prefix of k_transcript_step1 (kernels.hip) with its two stretches timed: instance absorbed / challenges drawn
template <bool EXCL> __global__ __launch_bounds__(64) void k_prefix(const uint8_t* inst_comp, const uint8_t* m_comp, int nproofs, int ell, uint64_t* out_state, Fr* out_vec_a, uint64_t* tsum) {
  __shared__ uint8_t scratch[64];
  const int p = blockIdx.x;
  if (p >= nproofs) return;
  if (EXCL) {   // the whole register file of the SIMD: no other wave can be placed beside this one
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
  }
  WaveStrobe t;
  t.set_lane(threadIdx.x);
  const uint64_t t0 = wall_clock64();
  t.init("curdleproofs", 12, scratch);
  for (int v = 0; v < 4; v++) {
    t.append_begin("curdleproofs_step1", 18, 8 + 48 * (size_t)ell, scratch);
    if (t.l < 8) scratch[8 + t.l] = (uint8_t)((uint64_t)ell >> (8 * t.l));
    wave_lds_sync();
    t.absorb(scratch + 8, 8);
    t.absorb(inst_comp + ((size_t)p * 4 + v) * ell * 48, 48 * (size_t)ell);
  }
  t.append_begin("curdleproofs_step1", 18, 48, scratch);
  t.absorb(m_comp + (size_t)p * 48, 48);
  const uint64_t t1 = wall_clock64();
  for (int i = 0; i < ell; i++) {
    Fr c;
    while (!t.challenge_attempt("curdleproofs_vec_a", 18, scratch, c)) {
    }
    if (threadIdx.x == 0) out_vec_a[(size_t)p * ell + i] = c;
  }
  const uint64_t t2 = wall_clock64();
  t.store(out_state + (size_t)p * 27);
  if (threadIdx.x == 0) { tsum[2 * p] = t1 - t0; tsum[2 * p + 1] = t2 - t1; }
}
int main() {
  {   // one permutation of a non-trivial state against the host's keccak_f1600, an absorb against the host's Strobe (strobe.hpp)
    uint64_t h[27], ref[25], got[58 + 27 + 12];
    for (int i = 0; i < 25; i++) h[i] = ref[i] = 0x9e3779b97f4a7c15ULL * (i + 1) ^ (0x0123456789abcdefULL << (i % 7));
    h[25] = 17; h[26] = 3;
    keccak_f1600(ref);
    uint8_t msg[1000];
    for (int i = 0; i < 1000; i++) msg[i] = (uint8_t)(i * 131 + 7);
    Strobe hs;
    for (int i = 0; i < 25; i++) hs.st[i] = h[i];
    hs.pos = 17; hs.pos_begin = 3;
    hs.absorb(msg, 1000);
    hs.xor_byte(hs.pos, 0xa7);
    Strobe hm;   // the Merlin run on the host
    for (int i = 0; i < 25; i++) hm.st[i] = h[i];
    hm.pos = 17; hm.pos_begin = 3;
    uint64_t chal[12];
    for (int rep = 0; rep < 3; rep++) {
      hm.append_message("some_label", msg, 48);
      uint32_t xv[8];
      for (int j = 0; j < 8; j++) xv[j] = 0x01234567u * (j + 1 + rep);
      xv[7] &= 0x0fffffffu;
      uint8_t xb[32];
      for (int j = 0; j < 32; j++) xb[j] = (uint8_t)(xv[j >> 2] >> (8 * (j & 3)));
      hm.append_message("a_scalar", xb, 32);
      uint32_t e[8];
      hm.challenge_scalar_canonical("a_challenge", e);
      Fr ce;
      for (int j = 0; j < 8; j++) ce.v[j] = e[j];
      const Fr cm = fe_to_mont(ce);
      for (int j = 0; j < 4; j++) chal[4 * rep + j] = (uint64_t)cm.v[2 * j] | ((uint64_t)cm.v[2 * j + 1] << 32);
    }
    uint64_t *din, *dout; uint8_t* dm;
    HIPCHECK(hipMalloc(&din, sizeof h)); HIPCHECK(hipMalloc(&dout, sizeof got)); HIPCHECK(hipMalloc(&dm, 1000));
    HIPCHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dm, msg, 1000, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, din, dout, dm, (size_t)1000);
    HIPCHECK(hipMemcpy(got, dout, sizeof got, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 25; i++) bad += got[i] != ref[i];
    bad += got[25] != 17 || got[26] != 3;
    int bad2 = 0;
    for (int i = 0; i < 25; i++) bad2 += got[27 + i] != hs.st[i];
    bad2 += got[27 + 25] != hs.pos || got[27 + 26] != hs.pos_begin;
    for (int j = 0; j < 4; j++) bad2 += got[54 + j] != hs.st[j];
    int bad3 = 0;
    for (int i = 0; i < 25; i++) bad3 += got[58 + i] != hm.st[i];
    bad3 += got[58 + 25] != hm.pos || got[58 + 26] != hm.pos_begin;
    for (int j = 0; j < 12; j++) bad3 += got[58 + 27 + j] != chal[j];
    printf("lane-parallel keccak-f[1600] against the host permutation: %d mismatches; absorb / xor_byte / peek4 against the host's Strobe: %d mismatches; "
           "append_message / append_scalar / challenge_scalar x 3: %d mismatches\n", bad, bad2, bad3);
    if (bad || bad2 || bad3) return 1;
  }
  const size_t len = 166 * 256;
  uint8_t* dmsg; uint64_t *dt, *dout;
  HIPCHECK(hipMalloc(&dmsg, len)); HIPCHECK(hipMemset(dmsg, 0x5a, len)); HIPCHECK(hipMalloc(&dt, 64)); HIPCHECK(hipMalloc(&dout, 64 * 8 * 4096));
  const int reps = 256;
  for (int blocks : {1, 64, 2048, 4096}) {
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, dmsg, len, dt, dout, reps);
      HIPCHECK(hipDeviceSynchronize());
    }
    uint64_t t[2];
    HIPCHECK(hipMemcpy(t, dt, sizeof t, hipMemcpyDeviceToHost));
    printf("%5d work-groups: keccak-f %.2f us each; absorbing a 166-byte block (+ its permutation) %.2f us\n", blocks, t[0] / 100.0 / reps, t[1] / 100.0 / 256);
  }
  {
    uint64_t h[27];
    for (int i = 0; i < 25; i++) h[i] = 0x9e3779b97f4a7c15ULL * (i + 1);
    h[25] = 17; h[26] = 3;
    uint64_t *din, *dts; Fr* dfr;
    HIPCHECK(hipMalloc(&din, sizeof h)); HIPCHECK(hipMalloc(&dts, 64 * 8)); HIPCHECK(hipMalloc(&dfr, 252 * sizeof(Fr)));
    HIPCHECK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_sections, dim3(1), dim3(64), 0, 0, din, dts, dfr, 252);
      HIPCHECK(hipDeviceSynchronize());
    }
    uint64_t ts[6];
    HIPCHECK(hipMemcpy(ts, dts, sizeof ts, hipMemcpyDeviceToHost));
    const char* names[6] = {"meta-AD(label || len)", "PRF header + forced permutation", "squeeze (peek4 + join)", "append_begin (label || len, AD header)", "absorb_words8", "fe_to_mont + store"};
    double tot = 0;
    for (int j = 0; j < 6; j++) tot += ts[j] / 100.0 / 252;
    printf("one challenge of the vec_a loop, alone on the GPU: %.2f us\n", tot);
    for (int j = 0; j < 6; j++) printf("  %-42s %.2f us\n", names[j], ts[j] / 100.0 / 252);
  }
  {
    const int np = 128, ell = 252;
    uint8_t *dinst, *dm; uint64_t *dst, *dts; Fr* dva;
    HIPCHECK(hipMalloc(&dinst, (size_t)np * 4 * ell * 48)); HIPCHECK(hipMalloc(&dm, np * 48)); HIPCHECK(hipMalloc(&dst, np * 27 * 8)); HIPCHECK(hipMalloc(&dts, np * 16));
    HIPCHECK(hipMalloc(&dva, (size_t)np * ell * sizeof(Fr)));
    HIPCHECK(hipMemset(dinst, 0x3c, (size_t)np * 4 * ell * 48)); HIPCHECK(hipMemset(dm, 0x11, np * 48));
    for (int excl = 0; excl < 2; excl++) {
      for (int rep = 0; rep < 2; rep++) {
        if (excl) hipLaunchKernelGGL(k_prefix<true>, dim3(np), dim3(64), 0, 0, dinst, dm, np, ell, dst, dva, dts);
        else hipLaunchKernelGGL(k_prefix<false>, dim3(np), dim3(64), 0, 0, dinst, dm, np, ell, dst, dva, dts);
        HIPCHECK(hipDeviceSynchronize());
      }
      uint64_t ts[2 * 128];
      HIPCHECK(hipMemcpy(ts, dts, sizeof ts, hipMemcpyDeviceToHost));
      double a = 0, b = 0, mx = 0;
      for (int i = 0; i < np; i++) { a += ts[2 * i] / 100.0; b += ts[2 * i + 1] / 100.0; mx = fmax(mx, (ts[2 * i] + ts[2 * i + 1]) / 100.0); }
      printf("transcript prefix, 128 proofs of ell = 252 alone on the GPU%s: instance absorbed in %.0f us, 252 challenges in %.0f us (means over the waves), slowest wave %.0f us\n",
             excl ? " (a SIMD's whole register file per wave)" : "", a / np, b / np, mx);
    }
  }
  return 0;
}
