// The reduction programs of the latency kernels (k_reduce_sets_wave, kernels.hip) — product code, host + device constexpr.
// A program is a list of rounds; in a round quad q of the wave runs job[q]: entry[dst] = entry[a] + entry[b], or, in a doubling
// round, entry[dst] = 2 entry[a], over an array of points in LDS.  Kept apart from the kernels so that the CPU tests can
// interpret the very tables the GPU runs (tests/test_host_emul.py::test_quad_reduction_programs).
#pragma once
#include <cstdint>

namespace cpx {

struct QuadJob {
  uint8_t a, b, dst, on;
};
// Entry map (97 entries = 21.7 KB of LDS per wave: 7 waves per CU; the first layout kept every stage in fresh entries, 145 = 32.5 KB:
// 4 waves per CU): inputs 0..63; the column tree overwrites its own first operand (entries 0..31: the row tree, which runs in the
// rounds before, has read them by then), the row tree lives in 64..95, the bit sums T and the plain sums P in the inputs that are
// dead after the first level (32..47), one entry that always holds the identity.
constexpr int QP_ROW = 64, QP_COL = 0, QP_T = 32, QP_P = 44, QP_IDENT = 96, QP_ENTRIES = 97, QP_MAX_ROUNDS = 20;
struct QuadProgram {
  int nrounds;
  int result;
  uint8_t dbl[QP_MAX_ROUNDS];
  QuadJob job[QP_MAX_ROUNDS][16];
};
constexpr void qp_level(QuadProgram& p, const QuadJob* jobs, int n, bool dbl) {
  for (int base = 0; base < n; base += 16) {
    for (int q = 0; q < 16; q++) p.job[p.nrounds][q] = base + q < n ? jobs[base + q] : QuadJob{(uint8_t)QP_IDENT, (uint8_t)QP_IDENT, (uint8_t)QP_IDENT, 0};
    p.dbl[p.nrounds++] = dbl;
  }
}
constexpr QuadJob qp_job(int a, int b, int dst) { return QuadJob{(uint8_t)a, (uint8_t)b, (uint8_t)dst, 1}; }
constexpr QuadProgram qp_plain() {
  QuadProgram p{};
  QuadJob jobs[32]{};
  for (int m = 32; m >= 1; m >>= 1) {
    for (int j = 0; j < m; j++) jobs[j] = qp_job(j, j + m, j);
    qp_level(p, jobs, m, false);
  }
  p.result = 0;
  return p;
}
// element i (0..3) of the numbers 0..7 with bit k set
constexpr int qp_with_bit(int i, int k) { return ((i >> k) << (k + 1)) | (1 << k) | (i & ((1 << k) - 1)); }
constexpr QuadProgram qp_weighted(bool upper) {
  QuadProgram p{};
  QuadJob jobs[64]{};
  for (int j = 0; j < 32; j++) {   // rows: A[8a+b] + A[8a+b+4]; columns: A[8a+b] + A[8(a+4)+b]
    jobs[j] = qp_job(8 * (j >> 2) + (j & 3), 8 * (j >> 2) + (j & 3) + 4, QP_ROW + j);
    jobs[32 + j] = qp_job(8 * (j >> 3) + (j & 7), 8 * ((j >> 3) + 4) + (j & 7), QP_COL + j);
  }
  qp_level(p, jobs, 64, false);
  for (int j = 0; j < 16; j++) {
    const int r = QP_ROW + 4 * (j >> 1) + (j & 1), c = QP_COL + 8 * (j >> 3) + (j & 7);
    jobs[j] = qp_job(r, r + 2, r);
    jobs[16 + j] = qp_job(c, c + 16, c);
  }
  qp_level(p, jobs, 32, false);
  for (int j = 0; j < 8; j++) {    // R_a = ROW[4a], C_b = COL[b]
    jobs[j] = qp_job(QP_ROW + 4 * j, QP_ROW + 4 * j + 1, QP_ROW + 4 * j);
    jobs[8 + j] = qp_job(QP_COL + j, QP_COL + 8 + j, QP_COL + j);
  }
  qp_level(p, jobs, 16, false);
  // bit sums: T[2g] = sum of the four R_a with bit g of a set (g < 3), of the four C_(w-1) with bit g - 3 of w set (w = 1..7)
  int n = 0;
  for (int g = 0; g < 6; g++)
    for (int h = 0; h < 2; h++) {
      const int k = g % 3, e0 = qp_with_bit(2 * h, k), e1 = qp_with_bit(2 * h + 1, k);
      jobs[n++] = g < 3 ? qp_job(QP_ROW + 4 * e0, QP_ROW + 4 * e1, QP_T + 2 * g + h) : qp_job(QP_COL + e0 - 1, QP_COL + e1 - 1, QP_T + 2 * g + h);
    }
  if (upper) for (int a = 0; a < 4; a++) jobs[n++] = qp_job(QP_ROW + 4 * a, QP_ROW + 4 * (a + 4), QP_P + a);   // the plain sum
  qp_level(p, jobs, n, false);
  n = 0;
  for (int g = 0; g < 6; g++) jobs[n++] = qp_job(QP_T + 2 * g, QP_T + 2 * g + 1, QP_T + 2 * g);
  if (upper) for (int a = 0; a < 2; a++) jobs[n++] = qp_job(QP_P + a, QP_P + a + 2, QP_P + a);
  qp_level(p, jobs, n, false);
  n = 0;
  jobs[n++] = qp_job(QP_T + 0, QP_COL + 7, QP_T + 0);   // weight 8: R-bit 0 and C_7
  if (upper) jobs[n++] = qp_job(QP_P + 0, QP_P + 1, QP_P + 0);
  qp_level(p, jobs, n, false);
  // weights 2^0 .. 2^6 at T[6], T[8], T[10], T[0], T[2], T[4], P[0]
  const int loc[7] = {QP_T + 6, QP_T + 8, QP_T + 10, QP_T + 0, QP_T + 2, QP_T + 4, QP_P + 0};
  const int nitems = upper ? 7 : 6;
  for (int sft = 0; sft < nitems - 1; sft++) {
    n = 0;
    for (int w = sft + 1; w < nitems; w++) jobs[n++] = qp_job(loc[w], loc[w], loc[w]);
    qp_level(p, jobs, n, true);
  }
  n = 0;
  jobs[n++] = qp_job(loc[0], loc[1], loc[0]);
  jobs[n++] = qp_job(loc[2], loc[3], loc[2]);
  jobs[n++] = qp_job(loc[4], loc[5], loc[4]);
  qp_level(p, jobs, n, false);
  n = 0;
  jobs[n++] = qp_job(loc[0], loc[2], loc[0]);
  if (upper) jobs[n++] = qp_job(loc[4], loc[6], loc[4]);
  qp_level(p, jobs, n, false);
  jobs[0] = qp_job(loc[0], loc[4], loc[0]);
  qp_level(p, jobs, 1, false);
  p.result = loc[0];
  return p;
}

}  // namespace cpx
