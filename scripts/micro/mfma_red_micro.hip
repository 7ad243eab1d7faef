// Can the matrix cores take the constant-multiplier half of a Montgomery reduction off the VALU?
//
// A product of the 28-bit-limb field (fp28.hpp) is 392 multiply-adds, half of which multiply the reduction digits m_0..m_13 by the
// CONSTANT p (f28_mul_body: acc += m[i] * P[k - i]).  "m * p for many m at once" is a matrix product with a Toeplitz matrix of p's
// digits, so in principle the int8 MFMA (v_mfma_i32_32x32x32_i8, ~4.4 POP/s on MI355X) could do it while the VALU does the a * b half.
// This micro-benchmark measures what that costs END TO END for one wave = 64 field elements (one per lane, as in every kernel here):
//
//   VALU   U = m * p as 27 columns of 64-bit sums, 196 v_mad_u64_u32 (p in SGPRs), then the carry pass to 28-bit limbs.
//   MFMA   1. re-layout: the 14 limbs of m -> 49 bytes -> signed int8 digits (xor 0x80; the bias 128 * sum 256^i * p is a constant that
//             the limb columns start from), 13 words per lane;
//          2. operands: the elements are the N dimension (B operand = digits), two batches of 32 elements per wave; ONE
//             v_permlane32_swap per digit register builds the B operands of both batches (lanes 32..63 must supply bytes 16..31 of the
//             element in lane l - 32);
//          3. 12 x v_mfma_i32_32x32x32_i8 (the 6 non-zero 32 x 32 blocks of the 128 x 64 Toeplitz band, per batch);
//          4. the 32 x 32 results come back with the rows of an element split over lanes l and l + 32: one v_permlane32_swap per
//             accumulator register (64 of them) brings every element's 98 byte-position sums c_j into its own lane;
//          5. recombination: U = sum_j c_j 256^j; byte position j goes to limb column floor(8 j / 28) with the multiplier
//             2^(8 j - 28 k): one v_mad_i64_i32 per c_j — 98 multiply-adds;  then the same carry pass.
// Both variants run the same recurrence (m' = low limbs of U xor high limbs of U) for `iters` rounds and must end in the same
// checksum.  The verdict is the ratio of the two times; note that a real reduction would ALSO need m itself (m = T_lo * (-p^-1) mod
// 2^392, a second constant product with its own re-layout) before this step — the measurement below is the cheaper half of the idea.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_red_micro.hip -o scripts/micro/mfma_red_micro
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                  \
  do {                                                         \
    hipError_t e_ = (x);                                       \
    if (e_ != hipSuccess) {                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                 \
    }                                                          \
  } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

static const uint32_t P28[14] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2,
                                 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x001a011};
__constant__ uint32_t c_p28[14];
constexpr uint32_t MASK28 = 0x0fffffffu;

__constant__ long long c_bias_col[28];   // 128 * sum_{i < 49} 256^i * p as 28-bit limbs: what the biased digits leave out
// carry pass: 28 columns (weight 2^(28 k)) -> 28 limbs of 28 bits
__device__ __forceinline__ void normalize(const int64_t (&col)[28], uint32_t (&limb)[28]) {
  int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 28; k++) {
    c += col[k];
    limb[k] = (uint32_t)c & MASK28;
    c >>= 28;
  }
}
__device__ __forceinline__ void next_m(const uint32_t (&limb)[28], uint32_t (&m)[14]) {
#pragma unroll
  for (int i = 0; i < 14; i++) m[i] = (limb[i] ^ limb[i + 14] ^ (limb[(i + 5) % 14] >> 3)) & MASK28;
}

// ---- VALU: 196 multiply-adds, p in scalar registers
__global__ __launch_bounds__(64) void k_valu(const uint32_t* __restrict__ in, int iters, uint32_t* __restrict__ out) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  uint32_t m[14];
#pragma unroll
  for (int i = 0; i < 14; i++) m[i] = in[g * 14 + i] & MASK28;
  uint32_t limb[28];
  for (int it = 0; it < iters; it++) {
    int64_t col[28];
    col[27] = 0;
#pragma unroll
    for (int k = 0; k < 27; k++) {
      uint64_t acc = 0;
#pragma unroll
      for (int i = 0; i < 14; i++)
        if (k - i >= 0 && k - i < 14) acc += (uint64_t)m[i] * c_p28[k - i];
      col[k] = (int64_t)acc;
    }
    normalize(col, limb);
    next_m(limb, m);
  }
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 28; i++) x = x * 31 + limb[i];
  out[g] = x;
}

// ---- MFMA: int8 Toeplitz blocks.  a_frag: [6 blocks][64 lanes] v4i = the A operand (row j of the block = lane & 31, k = 16 (lane >> 5)
// + byte index) of the non-zero blocks in the order (jb, kb) = (0,0) (1,0) (1,1) (2,0) (2,1) (3,1).
__global__ __launch_bounds__(64) void k_mfma(const uint32_t* __restrict__ in, const v4i* __restrict__ a_frag, int iters, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x, g = blockIdx.x * 64 + lane;
  uint32_t m[14];
#pragma unroll
  for (int i = 0; i < 14; i++) m[i] = in[g * 14 + i] & MASK28;
  v4i A[6];
#pragma unroll
  for (int b = 0; b < 6; b++) A[b] = a_frag[b * 64 + lane];
  uint32_t limb[28];
  for (int it = 0; it < iters; it++) {
    // 1. 14 limbs -> 13 words of the 392-bit string, bytes biased by 128
    uint32_t w[16];
#pragma unroll
    for (int t = 0; t < 13; t++) {
      const int bit = 32 * t, i = bit / 28, o = bit % 28;   // word t = bits [32 t, 32 t + 32)
      uint64_t x = (uint64_t)m[i] >> o;
      if (i + 1 < 14) x |= (uint64_t)m[i + 1] << (28 - o);
      if (i + 2 < 14 && 56 - o < 32) x |= (uint64_t)m[i + 2] << (56 - o);
      w[t] = (uint32_t)x;
    }
    w[12] &= 0xffu;   // bits 384..391: one byte
#pragma unroll
    for (int t = 0; t < 12; t++) w[t] ^= 0x80808080u;
    w[12] ^= 0x80u;   // byte 48 biased, bytes 49 .. 51 = 0
    w[13] = w[14] = w[15] = 0;
    // 2. B operands of both batches: k-step 0 = bytes 0..31 (words 0..7), k-step 1 = bytes 32..63 (words 8..15)
    v4i B0[2], B1[2];   // [k-step], batch 0 / batch 1
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const auto r = __builtin_amdgcn_permlane32_swap(w[8 * ks + q], w[8 * ks + 4 + q], false, false);
        B0[ks][q] = (int)r[0];   // lanes 0..31: own bytes 16 ks' .. (own element), lanes 32..63: bytes 16..31 of element lane - 32
        B1[ks][q] = (int)r[1];   // lanes 0..31: bytes 0..15 of element lane + 32, lanes 32..63: own bytes 16..31
      }
    }
    // 3. the six non-zero blocks per batch
    v16i D0[4], D1[4];
#pragma unroll
    for (int jb = 0; jb < 4; jb++) {
      D0[jb] = (v16i)0;
      D1[jb] = (v16i)0;
    }
    D0[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], B0[0], D0[0], 0, 0, 0);
    D1[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], B1[0], D1[0], 0, 0, 0);
    D0[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1], B0[0], D0[1], 0, 0, 0);
    D1[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1], B1[0], D1[1], 0, 0, 0);
    D0[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[2], B0[1], D0[1], 0, 0, 0);
    D1[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[2], B1[1], D1[1], 0, 0, 0);
    D0[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[3], B0[0], D0[2], 0, 0, 0);
    D1[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[3], B1[0], D1[2], 0, 0, 0);
    D0[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[4], B0[1], D0[2], 0, 0, 0);
    D1[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[4], B1[1], D1[2], 0, 0, 0);
    D0[3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[5], B0[1], D0[3], 0, 0, 0);
    D1[3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[5], B1[1], D1[3], 0, 0, 0);
    // 4. + 5. every element's byte-position sums into its own lane, then into the limb columns.  After the swap register r of D0 holds
    // row 32 jb + (r & 3) + 8 (r >> 2) of the own element, register r of D1 the row 4 further down.
    int64_t col[28];
#pragma unroll
    for (int k = 0; k < 28; k++) col[k] = c_bias_col[k];
#pragma unroll
    for (int jb = 0; jb < 4; jb++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int j0 = 32 * jb + (r & 3) + 8 * (r >> 2), j1 = j0 + 4;
        if (j0 > 96) continue;   // rows past the product's last byte are zero
        const auto s = __builtin_amdgcn_permlane32_swap((uint32_t)D0[jb][r], (uint32_t)D1[jb][r], false, false);
        const int c0 = (int)s[0], c1 = (int)s[1];
        {
          const int k = 8 * j0 / 28;
          col[k] += (int64_t)c0 * (int64_t)(1 << (8 * j0 - 28 * k));   // shift < 28: one v_mad_i64_i32
        }
        if (j1 <= 96) {
          const int k = 8 * j1 / 28;
          col[k] += (int64_t)c1 * (int64_t)(1 << (8 * j1 - 28 * k));
        }
      }
    }
    normalize(col, limb);
    next_m(limb, m);
  }
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 28; i++) x = x * 31 + limb[i];
  out[g] = x;
}

// signed base-256 digits (in -128 .. 127) of a non-negative integer given as little-endian bytes
static std::vector<int> signed_digits(const std::vector<uint32_t>& bytes) {
  std::vector<int> d;
  int carry = 0;
  for (size_t i = 0; i < bytes.size() || carry; i++) {
    int v = (i < bytes.size() ? (int)bytes[i] : 0) + carry;
    carry = 0;
    if (v >= 128) {
      v -= 256;
      carry = 1;
    }
    d.push_back(v);
  }
  return d;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 4 * 2;   // 2 waves per SIMD, like the MSM kernels
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_p28), P28, sizeof P28));
  // p as bytes; K = 128 * sum_{i < 49} 256^i; the Toeplitz matrix T[j][i] = ps[j - i] (i < 49) of p's signed digits
  std::vector<uint32_t> pb(49, 0);
  for (int bit = 0; bit < 14 * 28; bit++)
    if ((P28[bit / 28] >> (bit % 28)) & 1) pb[bit / 8] |= 1u << (bit % 8);
  while (!pb.empty() && pb.back() == 0) pb.pop_back();
  const std::vector<int> ps = signed_digits(pb);
  std::vector<uint32_t> kp(110, 0);   // K * p = 128 * sum_i (p << 8 i)
  {
    std::vector<uint64_t> acc(110, 0);
    for (int i = 0; i < 49; i++)
      for (size_t t = 0; t < pb.size(); t++) acc[i + t] += (uint64_t)pb[t] * 128u;
    uint64_t c = 0;
    for (size_t i = 0; i < acc.size(); i++) {
      c += acc[i];
      kp[i] = (uint32_t)(c & 255);
      c >>= 8;
    }
  }
  {
    long long bias[28] = {0};
    for (int bit = 0; bit < 110 * 8; bit++)
      if ((kp[bit / 8] >> (bit % 8)) & 1) {
        const int k = bit / 28 > 27 ? 27 : bit / 28;
        bias[k] += 1ll << (bit - 28 * k);
      }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(c_bias_col), bias, sizeof bias));
  }
  auto T = [&](int j, int i) -> int { return (i < 49 && j - i >= 0 && j - i < (int)ps.size()) ? ps[j - i] : 0; };
  const int blk[6][2] = {{0, 0}, {1, 0}, {1, 1}, {2, 0}, {2, 1}, {3, 1}};
  // blocks outside the list must be zero, and no row past 98 may be non-zero (the kernel skips them)
  for (int j = 0; j < 128; j++)
    for (int i = 0; i < 64; i++) {
      bool listed = false;
      for (auto& b : blk) listed |= b[0] == j / 32 && b[1] == i / 32;
      if ((!listed || j > 96) && T(j, i)) {
        fprintf(stderr, "Toeplitz entry (%d, %d) outside the computed band\n", j, i);
        return 1;
      }
    }
  std::vector<uint32_t> afr(6 * 64 * 4, 0);
  for (int b = 0; b < 6; b++)
    for (int l = 0; l < 64; l++)
      for (int q = 0; q < 16; q++) {
        const int j = 32 * blk[b][0] + (l & 31), i = 32 * blk[b][1] + 16 * (l >> 5) + q;
        afr[(b * 64 + l) * 4 + q / 4] |= (uint32_t)(uint8_t)(int8_t)T(j, i) << (8 * (q % 4));
      }
  const size_t n = (size_t)blocks * 64;
  std::vector<uint32_t> hin(n * 14);
  uint32_t s = 12345;
  for (auto& v : hin) {
    s = s * 1664525u + 1013904223u;
    v = s >> 3;
  }
  uint32_t *d_in, *d_o1, *d_o2;
  v4i* d_a;
  CK(hipMalloc(&d_in, hin.size() * 4));
  CK(hipMalloc(&d_o1, n * 4));
  CK(hipMalloc(&d_o2, n * 4));
  CK(hipMalloc(&d_a, afr.size() * 4));
  CK(hipMemcpy(d_in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_a, afr.data(), afr.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float t_valu = 1e30f, t_mfma = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    float ms;
    CK(hipEventRecord(a));
    k_valu<<<blocks, 64>>>(d_in, iters, d_o1);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    t_valu = ms < t_valu ? ms : t_valu;
    CK(hipEventRecord(a));
    k_mfma<<<blocks, 64>>>(d_in, d_a, iters, d_o2);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    t_mfma = ms < t_mfma ? ms : t_mfma;
  }
  std::vector<uint32_t> o1(n), o2(n);
  CK(hipMemcpy(o1.data(), d_o1, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o2.data(), d_o2, n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < n; i++) bad += o1[i] != o2[i];
  const double prods = (double)n * iters;
  printf("%s: %d waves (2 per SIMD) x 64 elements x %d rounds of  U = m * p  (392-bit m, 381-bit constant p) + carry pass\n", prop.name, blocks, iters);
  printf("VALU  196 v_mad_u64_u32 per element                          %8.3f ms  %7.2f G (m * p)/s\n", t_valu, prods / t_valu / 1e6);
  printf("MFMA  12 x v_mfma_i32_32x32x32_i8 per wave + re-layout        %8.3f ms  %7.2f G (m * p)/s   %s\n", t_mfma, prods / t_mfma / 1e6,
         bad ? "RESULTS DIFFER" : "same results as the VALU path");
  printf("ratio MFMA path / VALU path: %.2fx the time (%zu of %zu lanes differ)\n", t_mfma / t_valu, bad, n);
  return bad ? 1 : 0;
}
