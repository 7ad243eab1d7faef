// Scalar decomposition for the G1 endomorphism phi(x, y) = (beta x, y) — product code, host + device.
//
// BLS12-381: r = z^4 - z^2 + 1 with z = -0xd201000000010000, and phi acts on G1 as multiplication by -z^2
// (fp28.hpp: BETA; the same map the subgroup test of k_decompress uses).  With N = z^2 (128 bits) every scalar
// splits by ONE integer division:   k' = q N + t,  |t| <= N / 2,   k' P = t P + q (N P) = t P + q (-phi(P)).
// Taking k' = min(k, r - k) (and the sign with it) keeps q <= r / (2 N) + 1 < 2^127 and |t| < 2^127, so both halves
// have 16 signed radix-256 digits with no carry out of the top digit: a 32-copy table row is 16 doubling steps of
// 8 bits (120 doublings instead of 248) plus the 16 endomorphism images (one field product each), and the table
// kernel still sees 32 windows of weight 1.
#pragma once
#include <cstdint>
#include "mont32.hpp"

namespace cpx {

struct GlvCfg {
  static constexpr uint32_t N[4] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u};      // z^2
  static constexpr uint32_t M[4] = {0x7b67f717u, 0xb1fb7291u, 0xf00fd56eu, 0xbe35f678u};      // floor(2^255 / N)
  static constexpr uint32_t H[4] = {0x80000000u, 0x00000000u, 0x8000d201u, 0x5622d200u};      // floor(N / 2)
  static constexpr uint32_t HR[8] = {0x80000000u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};   // (r - 1) / 2
};

// k: canonical scalar (8 words, < r).  Outputs |t| and q (4 words each, < 2^127) with
//   k = (neg_k ? -1 : 1) * ((neg_t ? -1 : 1) * |t| + q N)   (mod r).
CPX_HD void glv_split(const uint32_t* k, uint32_t* t_abs, uint32_t* q, uint32_t& neg_k, uint32_t& neg_t) {
  // k' = min(k, r - k)
  int cmp = 0;   // sign of k - (r - 1) / 2
  CPX_UNROLL for (int i = 7; i >= 0; i--)
    if (cmp == 0) cmp = k[i] > GlvCfg::HR[i] ? 1 : (k[i] < GlvCfg::HR[i] ? -1 : 0);
  const bool gt = cmp > 0;
  neg_k = gt ? 1u : 0u;
  uint32_t a[8];
  {
    uint64_t br = 0;
    CPX_UNROLL for (int i = 0; i < 8; i++) {
      const uint64_t d = (uint64_t)FrCfg::P[i] - k[i] - br;
      const uint32_t rk = (uint32_t)d;
      br = (d >> 32) & 1;
      a[i] = gt ? rk : k[i];
    }
  }
  // A = k' + floor(N / 2)
  {
    uint64_t c = 0;
    CPX_UNROLL for (int i = 0; i < 8; i++) {
      c += (uint64_t)a[i] + (i < 4 ? GlvCfg::H[i] : 0u);
      a[i] = (uint32_t)c;
      c >>= 32;
    }
  }
  // q_est = floor(A M / 2^255)  (A < 2^254, so q_est is floor(A / N) or one less)
  uint32_t prod[12];
  CPX_UNROLL for (int i = 0; i < 12; i++) prod[i] = 0;
  CPX_UNROLL for (int j = 0; j < 4; j++) {
    uint64_t c = 0;
    CPX_UNROLL for (int i = 0; i < 8; i++) {
      c += (uint64_t)a[i] * GlvCfg::M[j] + prod[i + j];
      prod[i + j] = (uint32_t)c;
      c >>= 32;
    }
    prod[8 + j] = (uint32_t)c;
  }
  CPX_UNROLL for (int i = 0; i < 4; i++) q[i] = (prod[7 + i] >> 31) | (prod[8 + i] << 1);
  // rem = A - q N  in [0, 2 N): 5 words
  uint32_t qn[8];
  CPX_UNROLL for (int i = 0; i < 8; i++) qn[i] = 0;
  CPX_UNROLL for (int j = 0; j < 4; j++) {
    uint64_t c = 0;
    CPX_UNROLL for (int i = 0; i < 4; i++) {
      c += (uint64_t)q[i] * GlvCfg::N[j] + qn[i + j];
      qn[i + j] = (uint32_t)c;
      c >>= 32;
    }
    qn[4 + j] = (uint32_t)c;
  }
  uint32_t rem[5];
  {
    uint64_t br = 0;
    CPX_UNROLL for (int i = 0; i < 5; i++) {
      const uint64_t d = (uint64_t)a[i] - qn[i] - br;
      rem[i] = (uint32_t)d;
      br = (d >> 32) & 1;
    }
  }
  // if rem >= N: q += 1, rem -= N
  {
    uint32_t s[5];
    uint64_t br = 0;
    CPX_UNROLL for (int i = 0; i < 5; i++) {
      const uint64_t d = (uint64_t)rem[i] - (i < 4 ? GlvCfg::N[i] : 0u) - br;
      s[i] = (uint32_t)d;
      br = (d >> 32) & 1;
    }
    const bool ge = br == 0;
    uint64_t c = ge ? 1 : 0;
    CPX_UNROLL for (int i = 0; i < 4; i++) {
      c += q[i];
      q[i] = (uint32_t)c;
      c >>= 32;
    }
    CPX_UNROLL for (int i = 0; i < 5; i++) rem[i] = ge ? s[i] : rem[i];
  }
  // t = rem - floor(N / 2)  in [-N/2, N/2]
  {
    uint32_t d1[4], d2[4];
    uint64_t b1 = 0, b2 = 0;
    CPX_UNROLL for (int i = 0; i < 4; i++) {
      const uint64_t x = (uint64_t)rem[i] - GlvCfg::H[i] - b1;
      d1[i] = (uint32_t)x;
      b1 = (x >> 32) & 1;
      const uint64_t y = (uint64_t)GlvCfg::H[i] - rem[i] - b2;
      d2[i] = (uint32_t)y;
      b2 = (y >> 32) & 1;
    }
    const bool negt = b1 != 0;   // rem < H (rem < N < 2^128 here, so 4 words decide)
    neg_t = negt ? 1u : 0u;
    CPX_UNROLL for (int i = 0; i < 4; i++) t_abs[i] = negt ? d2[i] : d1[i];
  }
}

// Signed radix-256 digits of a value < 2^127 (4 words): byte w of v + 0x8080...80 is d_w + 128, d_w in [-128, 127],
// sum d_w 256^w = v; no carry leaves the top byte.
CPX_HD void glv_biased_bytes(const uint32_t* v, uint32_t* out) {
  uint64_t c = 0;
  CPX_UNROLL for (int i = 0; i < 4; i++) {
    c += (uint64_t)v[i] + 0x80808080u;
    out[i] = (uint32_t)c;
    c >>= 32;
  }
}

}  // namespace cpx
