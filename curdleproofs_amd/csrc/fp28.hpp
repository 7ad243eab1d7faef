// BLS12-381 Fp on 14 signed 28-bit limbs — the carry-free field of the table kernels (product code).
//
// Why: on gfx950 the carry instructions are as expensive as the multiplier (measured, profiles/r01_mac_micro.txt:
// v_mad_u64_u32 ~4.5 cycles per wave, v_addc_co_u32 ~4.5 cycles, both "half rate"), so a saturated 32-bit-limb
// Montgomery product costs ~9 cycles per limb product.  With 28-bit limbs a whole column of the product-scanning
// Montgomery multiplication (<= 28 limb products of < 2^56) accumulates in ONE 64-bit register with plain
// v_mad_i64_i32 and no carry handling at all: 392 multiply-accumulates at ~4.5-5.5 cycles instead of 288 at ~9.
//
// Representation: value = sum v[i] * 2^(28 i), v[0..12] in [0, 2^28) after normalisation, v[13] signed (the
// value itself may be negative).  Montgomery radix R = 2^392.  Values are LAZY: additions and subtractions are
// limb-wise with a carry pass and NO modular reduction; a product returns a value in (-0.81 p, 1.81 p) provided
// |a| * |b| < 2^11.3 * p^2, which the point formulas keep with a wide margin (worst case 38 p * 38 p, see g1_28).
// Exact zero is all-limbs-zero; a product is = 0 mod p iff it equals 0 or p (it cannot reach -p).
#pragma once
#include "mont32.hpp"
#include "modinv30.hpp"

namespace cpx {

struct F28 {
  int32_t v[14];
  static CPX_HD F28 zero() {
    F28 r;
    CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = 0;
    return r;
  }
  CPX_HD bool is_zero_exact() const {
    int32_t o = 0;
    CPX_UNROLL for (int i = 0; i < 14; i++) o |= v[i];
    return o == 0;
  }
};

struct F28Cfg {
  static constexpr int32_t MASK = 0x0fffffff;
  static constexpr int32_t P[14] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2,
                                    0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x001a011};
  static constexpr uint32_t INV = 0xffcfffd;   // -p^-1 mod 2^28
  // 2^392 mod p (Montgomery one), 2^400 mod p (standard -> internal), 2^384 mod p (internal -> standard)
  static constexpr int32_t ONE[14] = {0x347fcb8, 0xd800000, 0x002b119, 0x0cde6d2, 0xc7212e0, 0x83a2090, 0x037669f,
                                      0xda0f73e, 0x9b09b42, 0x1297bb0, 0x515d98f, 0x012ca7c, 0x659fcfa, 0x000577a};
  static constexpr int32_t C_IN[14] = {0x80e6299, 0x3500034, 0xeb12856, 0xdeb2699, 0xc988670, 0x4ef6697, 0x70983e8,
                                       0xa4e6fe9, 0x3e8a053, 0xecf271e, 0xc20d323, 0x6eb6385, 0x47f1286, 0x00156da};
  // 4 and the cube root of unity beta of the G1 endomorphism (x, y) -> (beta x, y) = [-u^2](x, y), Montgomery form
  static constexpr int32_t FOUR[14] = {0xd1ff2e0, 0x6000000, 0x00ac467, 0x3379b48, 0x1c84b80, 0x0e88243, 0x0dd9a7e,
                                       0x683dcf8, 0x6c26d0b, 0x4a5eec2, 0x457663c, 0x04b29f1, 0x967f3e8, 0x0015de9};
  static constexpr int32_t BETA[14] = {0xa75929a, 0x681b798, 0x22a3e9d, 0xabc02bf, 0x4e5bb45, 0x55e6e7e, 0x4814117,
                                       0x6d04f1b, 0xae3387d, 0x54acb0c, 0x0a4c74b, 0x56138b5, 0xb64e066, 0x00076f2};
  // 2^1176 mod p: plain integer -> Montgomery form with two extra factors of 2^392 (used after an integer inversion)
  static constexpr int32_t C_INV[14] = {0x1f7b890, 0x294cc4d, 0x9f3af22, 0xb5ba56c, 0xcb5c0cc, 0xc0d975c, 0xc89a8c5,
                                        0x6c968b4, 0x22672ea, 0x91de8c9, 0x35652a6, 0x84977c8, 0x424bbb9, 0x00141ab};
  static constexpr int32_t C_OUT[14] = {0x002fffd, 0x0900000, 0xc000276, 0x000bc40, 0x8baebf4, 0x5753c75, 0x55f4898,
                                        0x7052574, 0x7ce5853, 0x56ec6d7, 0x71a97a2, 0xe4935c0, 0xec3fa80, 0x0015f65};
};

// carry pass: limbs 0..12 into [0, 2^28), the top limb absorbs the (signed) rest
CPX_HD void f28_normalize(F28& a) {
  CPX_UNROLL for (int i = 0; i < 13; i++) {
    const int32_t c = a.v[i] >> 28;   // arithmetic shift
    a.v[i] &= F28Cfg::MASK;
    a.v[i + 1] += c;
  }
}
CPX_HD F28 f28_add(const F28& a, const F28& b) {
  F28 r;
  CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = a.v[i] + b.v[i];
  f28_normalize(r);
  return r;
}
CPX_HD F28 f28_sub(const F28& a, const F28& b) {
  F28 r;
  CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = a.v[i] - b.v[i];
  f28_normalize(r);
  return r;
}
// Differences that only FEED PRODUCTS skip the carry pass: the limb-wise difference of two normalised values has signed limbs
// below 2^28 in magnitude, the column sums of the signed multiply-adds keep the bounds of normalised operands (28 terms of
// < 2^56 per product), and the product's own output is normalised again.  Not for values that are stored, tested for zero or
// run through further additions (their limbs would grow): those take f28_sub.  (39 instructions per carry pass, ~5 % of a mixed
// addition's instructions in total.)
CPX_HD F28 f28_sub_lazy(const F28& a, const F28& b) {
  F28 r;
  CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = a.v[i] - b.v[i];
  return r;
}
CPX_HD F28 f28_cneg_lazy(const F28& a, bool neg) {   // +-a for a normalised a: limbs in (-2^28, 2^28)
  F28 r;
  CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = neg ? -a.v[i] : a.v[i];
  return r;
}
// a - b - 2 c with one carry pass (limbs stay below 2^30 before it)
CPX_HD F28 f28_sub_sub2(const F28& a, const F28& b, const F28& c) {
  F28 r;
  CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = a.v[i] - b.v[i] - 2 * c.v[i];
  f28_normalize(r);
  return r;
}
CPX_HD F28 f28_neg(const F28& a) {
  F28 r;
  CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = -a.v[i];
  f28_normalize(r);
  return r;
}
// a * 2^k for tiny k (limbs < 2^28 -> < 2^31 for k <= 3 before the carry pass; the top limb stays small)
template <int K> CPX_HD F28 f28_shl(const F28& a) {
  F28 r;
  CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = a.v[i] * (int32_t)(1 << K);   // (a multiplication: shifting a negative limb left is undefined before C++20; same instruction)
  f28_normalize(r);
  return r;
}
CPX_HD F28 f28_cneg(const F28& a, bool neg) {
  F28 n = f28_neg(a), r;
  CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = neg ? n.v[i] : a.v[i];
  return r;
}

// Montgomery product a * b / 2^392 mod p (lazy range).  Product scanning: column k gathers a_i b_(k-i) and
// m_i p_(k-i) into one signed 64-bit accumulator (|sum| < 2^62), emits one limb, shifts by 28 bits.
CPX_HD F28 f28_mul_body(const F28& a, const F28& b) {
  int32_t m[14];
  F28 t;
  int64_t acc = 0;
  CPX_UNROLL for (int k = 0; k < 14; k++) {
    CPX_UNROLL for (int i = 0; i < k; i++) {
      acc += (int64_t)a.v[i] * b.v[k - i];
      acc += (int64_t)m[i] * F28Cfg::P[k - i];
    }
    acc += (int64_t)a.v[k] * b.v[0];
    m[k] = (int32_t)(((uint32_t)acc * F28Cfg::INV) & (uint32_t)F28Cfg::MASK);
    acc += (int64_t)m[k] * F28Cfg::P[0];
    acc >>= 28;
  }
  CPX_UNROLL for (int k = 14; k < 27; k++) {
    CPX_UNROLL for (int i = k - 13; i < 14; i++) {
      acc += (int64_t)a.v[i] * b.v[k - i];
      acc += (int64_t)m[i] * F28Cfg::P[k - i];
    }
    t.v[k - 14] = (int32_t)acc & F28Cfg::MASK;
    acc >>= 28;
  }
  t.v[13] = (int32_t)acc;
  return t;
}
// a * b - c * d with ONE Montgomery reduction: column k gathers a_i b_(k-i) - c_i d_(k-i) and m_i p_(k-i) (84 terms, |sum| < 2^63
// for normalised limbs): 588 multiply-adds instead of the 784 of two products.  The mixed addition ends in such a difference
// (Y3 = R (Q - X3) - Y1 PPP).  Result in (-1.62 p - eps, 2.62 p) for operands at the bound of f28_mul_body, the same interval
// the difference of two reduced products spans; NOT a "product" for f28_product_is_zero.
CPX_HD F28 f28_mulsub_body(const F28& a, const F28& b, const F28& c, const F28& d) {
  int32_t m[14], nc[14];
  CPX_UNROLL for (int i = 0; i < 14; i++) nc[i] = -c.v[i];
  F28 t;
  int64_t acc = 0;
  CPX_UNROLL for (int k = 0; k < 14; k++) {
    CPX_UNROLL for (int i = 0; i < k; i++) {
      acc += (int64_t)a.v[i] * b.v[k - i];
      acc += (int64_t)nc[i] * d.v[k - i];
      acc += (int64_t)m[i] * F28Cfg::P[k - i];
    }
    acc += (int64_t)a.v[k] * b.v[0];
    acc += (int64_t)nc[k] * d.v[0];
    m[k] = (int32_t)(((uint32_t)acc * F28Cfg::INV) & (uint32_t)F28Cfg::MASK);
    acc += (int64_t)m[k] * F28Cfg::P[0];
    acc >>= 28;
  }
  CPX_UNROLL for (int k = 14; k < 27; k++) {
    CPX_UNROLL for (int i = k - 13; i < 14; i++) {
      acc += (int64_t)a.v[i] * b.v[k - i];
      acc += (int64_t)nc[i] * d.v[k - i];
      acc += (int64_t)m[i] * F28Cfg::P[k - i];
    }
    t.v[k - 14] = (int32_t)acc & F28Cfg::MASK;
    acc >>= 28;
  }
  t.v[13] = (int32_t)acc;
  return t;
}
// out-of-line entry with scalar register arguments (same calling-convention reasoning as fe_mul_regs12)
#define CPX_L14(p) p##0, p##1, p##2, p##3, p##4, p##5, p##6, p##7, p##8, p##9, p##10, p##11, p##12, p##13
#define CPX_A14(p) int32_t p##0, int32_t p##1, int32_t p##2, int32_t p##3, int32_t p##4, int32_t p##5, int32_t p##6, int32_t p##7, int32_t p##8, int32_t p##9, int32_t p##10, int32_t p##11, int32_t p##12, int32_t p##13
CPX_HD_FN F28 f28_mul_regs(CPX_A14(a), CPX_A14(b)) {
  const F28 x{{CPX_L14(a)}}, y{{CPX_L14(b)}};
  return f28_mul_body(x, y);
}
CPX_HD F28 f28_mul(const F28& a, const F28& b) {
  return f28_mul_regs(a.v[0], a.v[1], a.v[2], a.v[3], a.v[4], a.v[5], a.v[6], a.v[7], a.v[8], a.v[9], a.v[10], a.v[11], a.v[12], a.v[13],
                      b.v[0], b.v[1], b.v[2], b.v[3], b.v[4], b.v[5], b.v[6], b.v[7], b.v[8], b.v[9], b.v[10], b.v[11], b.v[12], b.v[13]);
}
// Squaring: the 91 off-diagonal limb products are taken once against the doubled operand (301 multiply-adds
// instead of 392).
CPX_HD F28 f28_sqr_body(const F28& a) {
  int32_t m[14], d[14];
  CPX_UNROLL for (int i = 0; i < 14; i++) d[i] = a.v[i] * 2;   // |limb| < 2^29
  F28 t;
  int64_t acc = 0;
  CPX_UNROLL for (int k = 0; k < 14; k++) {
    CPX_UNROLL for (int i = 0; 2 * i < k; i++) acc += (int64_t)d[i] * a.v[k - i];
    if ((k & 1) == 0) acc += (int64_t)a.v[k / 2] * a.v[k / 2];
    CPX_UNROLL for (int i = 0; i < k; i++) acc += (int64_t)m[i] * F28Cfg::P[k - i];
    m[k] = (int32_t)(((uint32_t)acc * F28Cfg::INV) & (uint32_t)F28Cfg::MASK);
    acc += (int64_t)m[k] * F28Cfg::P[0];
    acc >>= 28;
  }
  CPX_UNROLL for (int k = 14; k < 27; k++) {
    CPX_UNROLL for (int i = k - 13; 2 * i < k; i++) acc += (int64_t)d[i] * a.v[k - i];
    if ((k & 1) == 0) acc += (int64_t)a.v[k / 2] * a.v[k / 2];
    CPX_UNROLL for (int i = k - 13; i < 14; i++) acc += (int64_t)m[i] * F28Cfg::P[k - i];
    t.v[k - 14] = (int32_t)acc & F28Cfg::MASK;
    acc >>= 28;
  }
  t.v[13] = (int32_t)acc;
  return t;
}
CPX_HD_FN F28 f28_sqr_regs(CPX_A14(a)) {
  const F28 x{{CPX_L14(a)}};
  return f28_sqr_body(x);
}
CPX_HD F28 f28_sqr(const F28& a) {
  return f28_sqr_regs(a.v[0], a.v[1], a.v[2], a.v[3], a.v[4], a.v[5], a.v[6], a.v[7], a.v[8], a.v[9], a.v[10], a.v[11], a.v[12], a.v[13]);
}

CPX_HD F28 f28_const(const int32_t* c) {
  F28 r;
  CPX_UNROLL for (int i = 0; i < 14; i++) r.v[i] = c[i];
  return r;
}
CPX_HD F28 f28_one() { return f28_const(F28Cfg::ONE); }

// a PRODUCT (value in (-0.81p, 1.81p), normalised limbs) is zero mod p iff it is 0 or p
CPX_HD bool f28_product_is_zero(const F28& a) {
  int32_t z = 0, e = 0;
  CPX_UNROLL for (int i = 0; i < 14; i++) {
    z |= a.v[i];
    e |= a.v[i] ^ F28Cfg::P[i];
  }
  return z == 0 || e == 0;
}

// standard wire form (12 x u32 Montgomery, R = 2^384, canonical) <-> internal
CPX_HD F28 f28_from_words(const uint32_t* w);
CPX_HD F28 f28_from_std(const Fp& s) {
  return f28_mul(f28_from_words(s.v), f28_const(F28Cfg::C_IN));   // X * 2^400 / 2^392 = X * 2^8 = x * 2^392
}
// canonical integer value (12 x u32) of a PRODUCT-range lazy value (-0.81p, 1.81p)
CPX_HD void f28_canonical_words(F28 t, uint32_t* w) {
  const bool neg = t.v[13] < 0;
  CPX_UNROLL for (int i = 0; i < 14; i++) t.v[i] += neg ? F28Cfg::P[i] : 0;
  f28_normalize(t);
  F28 d;
  CPX_UNROLL for (int i = 0; i < 14; i++) d.v[i] = t.v[i] - F28Cfg::P[i];
  f28_normalize(d);
  const bool ge = d.v[13] >= 0;
  CPX_UNROLL for (int i = 0; i < 14; i++) t.v[i] = ge ? d.v[i] : t.v[i];
  CPX_UNROLL for (int k = 0; k < 12; k++) {
    const int bit = 32 * k, i = bit / 28, o = bit % 28;
    uint64_t x = (uint64_t)(uint32_t)t.v[i] >> o;
    if (i + 1 < 14) x |= (uint64_t)(uint32_t)t.v[i + 1] << (28 - o);
    if (i + 2 < 14 && 56 - o < 32) x |= (uint64_t)(uint32_t)t.v[i + 2] << (56 - o);
    w[k] = (uint32_t)x;
  }
}
CPX_HD F28 f28_from_words(const uint32_t* w) {   // plain 28-bit limbs of a 384-bit integer (no Montgomery factor applied)
  F28 u;
  CPX_UNROLL for (int i = 0; i < 14; i++) {
    const int bit = 28 * i, k = bit >> 5, o = bit & 31;
    uint64_t x = w[k];
    if (k + 1 < 12) x |= (uint64_t)w[k + 1] << 32;
    u.v[i] = (int32_t)((x >> o) & (uint32_t)F28Cfg::MASK);
  }
  return u;
}
CPX_HD Fp f28_to_std(const F28& a) {
  Fp r;
  f28_canonical_words(f28_mul(a, f28_const(F28Cfg::C_OUT)), r.v);   // x * 2^384, canonical
  return r;
}
// Inverse of a lazy value by batched division steps (modinv30.hpp):
// X = a 2^392 -> X^-1 = a^-1 2^-392 as a plain integer -> one product with 2^1176 gives a^-1 2^392.
CPX_HD F28 f28_inv_euclid(const F28& a) {
  uint32_t w[12], iw[12];
  f28_canonical_words(f28_mul(a, f28_one()), w);   // any lazy input: one product brings it into the canonicalisable range
  words_inv_mod_p_divsteps(w, iw);
  return f28_mul(f28_from_words(iw), f28_const(F28Cfg::C_INV));
}

}  // namespace cpx
