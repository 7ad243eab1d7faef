// ORACLE (test infrastructure — never linked into the product path).
//
// CPU restatement of BLS12-381 G1 group arithmetic as the reference uses it via
// ark-ec ^0.4 / ark-bls12-381 ^0.4 (Cargo.toml:22,24; not vendored):
//   G1Affine{x,y,infinity}, G1Projective (Jacobian X,Y,Z), Mul<Fr> (MSB-first
//   double-and-add), into_affine / normalize_batch, VariableBaseMSM::msm (Pippenger),
//   zcash-format compressed (de)serialisation with subgroup check.
// Reference call sites: util.rs:19-29 (msm, msm_from_projective), util.rs:94-95,
// inner_product_argument.rs:174-179 and same_multiscalar_argument.rs:126-131 (folds),
// grand_product_argument.rs:90-102 (rescale), whisk.rs:313-320 (bytes <-> G1Affine).
// Any correct algorithm yields the same group element, hence the same bytes after
// normalisation; the algorithms below follow the reference's *choices* (per-element
// double-and-add, Pippenger with ark-ec's window rule) so this file doubles as the CPU baseline.
#pragma once
#include "field.h"
#include <vector>

namespace orc {

struct G1Aff {
  Fp x, y;
  bool inf;
  static G1Aff identity() { return G1Aff{Fp::zero(), Fp::zero(), true}; }
  bool operator==(const G1Aff& o) const {
    if (inf || o.inf) return inf == o.inf;
    return x == o.x && y == o.y;
  }
};

struct G1 {   // Jacobian: (X/Z^2, Y/Z^3); Z == 0 <=> identity
  Fp x, y, z;
  static G1 identity() { return G1{Fp::one(), Fp::one(), Fp::zero()}; }
  static G1 from_affine(const G1Aff& a) {
    if (a.inf) return identity();
    return G1{a.x, a.y, Fp::one()};
  }
  bool is_identity() const { return z.is_zero(); }
};

static inline Fp fp_b() { return Fp::from_u64(4); }   // y^2 = x^3 + 4

static inline G1 g1_double(const G1& p) {
  if (p.is_identity()) return p;
  // dbl-2009-l (a = 0)
  Fp A = p.x.sqr(), B = p.y.sqr(), C = B.sqr();
  Fp D = ((p.x + B).sqr() - A - C).dbl();
  Fp E = A + A + A, F = E.sqr();
  G1 r;
  r.x = F - D.dbl();
  r.y = E * (D - r.x) - C.dbl().dbl().dbl();
  r.z = (p.y * p.z).dbl();
  return r;
}

static inline G1 g1_add(const G1& p, const G1& q) {
  if (p.is_identity()) return q;
  if (q.is_identity()) return p;
  Fp z1z1 = p.z.sqr(), z2z2 = q.z.sqr();
  Fp u1 = p.x * z2z2, u2 = q.x * z1z1;
  Fp s1 = p.y * q.z * z2z2, s2 = q.y * p.z * z1z1;
  if (u1 == u2) {
    if (s1 == s2) return g1_double(p);
    return G1::identity();
  }
  Fp h = u2 - u1, i = h.dbl().sqr(), j = h * i, rr = (s2 - s1).dbl(), v = u1 * i;
  G1 r;
  r.x = rr.sqr() - j - v.dbl();
  r.y = rr * (v - r.x) - (s1 * j).dbl();
  r.z = ((p.z + q.z).sqr() - z1z1 - z2z2) * h;
  return r;
}

static inline G1 g1_add_mixed(const G1& p, const G1Aff& q) {
  if (q.inf) return p;
  if (p.is_identity()) return G1::from_affine(q);
  Fp z1z1 = p.z.sqr();
  Fp u2 = q.x * z1z1, s2 = q.y * p.z * z1z1;
  if (p.x == u2) {
    if (p.y == s2) return g1_double(p);
    return G1::identity();
  }
  Fp h = u2 - p.x, hh = h.sqr(), i = hh.dbl().dbl(), j = h * i, rr = (s2 - p.y).dbl(), v = p.x * i;
  G1 r;
  r.x = rr.sqr() - j - v.dbl();
  r.y = rr * (v - r.x) - (p.y * j).dbl();
  r.z = (p.z + h).sqr() - z1z1 - hh;
  return r;
}

static inline G1 g1_neg(const G1& p) { return G1{p.x, -p.y, p.z}; }
static inline G1Aff g1_neg(const G1Aff& p) { return G1Aff{p.x, -p.y, p.inf}; }
static inline G1 g1_sub(const G1& p, const G1& q) { return g1_add(p, g1_neg(q)); }

static inline bool g1_eq(const G1& p, const G1& q) {
  if (p.is_identity() || q.is_identity()) return p.is_identity() == q.is_identity();
  Fp z1z1 = p.z.sqr(), z2z2 = q.z.sqr();
  if (p.x * z2z2 != q.x * z1z1) return false;
  return p.y * (z2z2 * q.z) == q.y * (z1z1 * p.z);
}

// ark-ec `into_affine`: one field inversion per point
// (the per-element cost the reference pays at inner_product_argument.rs:177-178).
static inline G1Aff g1_to_affine(const G1& p) {
  if (p.is_identity()) return G1Aff::identity();
  Fp zi = p.z.inverse(), zi2 = zi.sqr();
  return G1Aff{p.x * zi2, p.y * zi2 * zi, false};
}

// ark-ec `normalize_batch` (util.rs:27): Montgomery's trick, identity-safe
static inline std::vector<G1Aff> g1_batch_to_affine(const std::vector<G1>& ps) {
  size_t n = ps.size();
  std::vector<Fp> pre(n);
  Fp acc = Fp::one();
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    if (!ps[i].is_identity()) acc = acc * ps[i].z;
  }
  Fp inv = acc.inverse();
  std::vector<G1Aff> out(n);
  for (size_t i = n; i-- > 0;) {
    if (ps[i].is_identity()) {
      out[i] = G1Aff::identity();
      continue;
    }
    Fp zi = inv * pre[i];
    inv = inv * ps[i].z;
    Fp zi2 = zi.sqr();
    out[i] = G1Aff{ps[i].x * zi2, ps[i].y * zi2 * zi, false};
  }
  return out;
}

// Scalar multiplication by a canonical little-endian integer, MSB-first double-and-add
// (ark-ec `mul_bigint` default, used by every `.mul(` in the reference).
static inline G1 g1_mul_limbs(const G1& p, const uint64_t* k, int nlimbs) {
  G1 r = G1::identity();
  bool started = false;
  for (int i = nlimbs * 64 - 1; i >= 0; i--) {
    if (started) r = g1_double(r);
    if ((k[i / 64] >> (i % 64)) & 1) {
      r = g1_add(r, p);
      started = true;
    }
  }
  return r;
}
static inline G1 g1_mul(const G1& p, const Fr& k) {
  uint64_t c[4];
  k.to_canonical(c);
  return g1_mul_limbs(p, c, 4);
}
static inline G1 g1_mul(const G1Aff& p, const Fr& k) { return g1_mul(G1::from_affine(p), k); }

static inline bool g1_on_curve(const G1Aff& p) {
  if (p.inf) return true;
  return p.y.sqr() == p.x.sqr() * p.x + fp_b();
}
static inline bool g1_in_subgroup(const G1Aff& p) {
  if (p.inf) return true;
  return g1_mul_limbs(G1::from_affine(p), FrParams::MOD, 4).is_identity();
}

static inline G1Aff g1_generator() {
  static const uint64_t gx[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                                 0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
  static const uint64_t gy[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                                 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};
  return G1Aff{Fp::from_canonical(gx), Fp::from_canonical(gy), false};
}

// ---- zcash compressed encoding (ark-bls12-381 g1 serialisation; whisk.rs:313-320) ----
// 48 bytes big-endian x; byte0 bit7 = compressed, bit6 = infinity, bit5 = (y > -y).
static inline void g1_compress(const G1Aff& p, uint8_t out[48]) {
  memset(out, 0, 48);
  if (p.inf) {
    out[0] = 0xc0;
    return;
  }
  uint64_t c[6];
  p.x.to_canonical(c);
  for (int i = 0; i < 48; i++) out[i] = (uint8_t)(c[5 - i / 8] >> (56 - 8 * (i % 8)));
  out[0] |= 0x80;
  if (p.y.cmp_canonical(-p.y) > 0) out[0] |= 0x20;
}
static inline void g1_compress(const G1& p, uint8_t out[48]) { g1_compress(g1_to_affine(p), out); }

// ark-ec `get_point_from_x_unchecked(x, greatest)`
static inline bool g1_point_from_x(const Fp& x, bool greatest, G1Aff* out) {
  Fp y;
  if (!fp_sqrt(x.sqr() * x + fp_b(), &y)) return false;
  Fp ny = -y;
  bool y_is_smaller = y.cmp_canonical(ny) < 0;
  Fp small = y_is_smaller ? y : ny, large = y_is_smaller ? ny : y;
  *out = G1Aff{x, greatest ? large : small, false};
  return true;
}

// Encodings with the infinity flag set.  ark-bls12-381 ^0.4 `read_g1_compressed` (curves/util.rs of 0.4.0, a crates.io dependency that is
// not vendored under /root/reference — recalled, not re-read): `if flags.is_infinity { return Ok(G1Affine::zero()); }` right after the
// compression-flag test, i.e. neither the sort flag nor the other bits are looked at; 0.5 rejects both.  strict = false restates 0.4,
// strict = true the zcash specification's wording (only 0xc0 || 0^47).  Process-wide switch for the tests (orc_set_strict_infinity).
static inline bool& g1_strict_infinity() {
  static bool strict = false;
  return strict;
}

// returns false on any malformed / off-curve / out-of-subgroup encoding
// (ark `deserialize_compressed` = Compress::Yes, Validate::Yes)
static inline bool g1_decompress(const uint8_t in[48], G1Aff* out, bool check_subgroup = true) {
  bool compressed = in[0] & 0x80, infinity = in[0] & 0x40, sort = in[0] & 0x20;
  if (!compressed) return false;
  uint8_t b[48];
  memcpy(b, in, 48);
  b[0] &= 0x1f;
  if (infinity) {
    if (g1_strict_infinity()) {
      if (sort) return false;
      for (int i = 0; i < 48; i++)
        if (b[i]) return false;
    }
    *out = G1Aff::identity();
    return true;
  }
  uint64_t c[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 48; i++) c[5 - i / 8] |= (uint64_t)b[i] << (56 - 8 * (i % 8));
  if (big_cmp<6>(c, FpParams::MOD) >= 0) return false;
  Fp x = Fp::from_canonical(c);
  if (!g1_point_from_x(x, sort, out)) return false;
  if (check_subgroup && !g1_in_subgroup(*out)) return false;
  return true;
}

// ---- Pippenger MSM, following ark-ec 0.4 VariableBaseMSM's structure (util.rs:19-22) ----
// window c = 3 if n < 32 else ln_without_floats(n) + 2, ln_without_floats(a) = log2(a)*69/100;
// per window: bucket accumulation with mixed adds, running-sum reduction; Horner over windows.
static inline unsigned msm_window_bits(size_t n) {
  if (n < 32) return 3;
  unsigned lg = 0;
  while ((size_t(1) << (lg + 1)) <= n) lg++;   // floor(log2 n)
  // ark_std::log2 is ceil(log2 n)
  unsigned clg = ((size_t(1) << lg) == n) ? lg : lg + 1;
  return clg * 69 / 100 + 2;
}

static inline G1 g1_msm(const G1Aff* bases, const Fr* scalars, size_t n) {
  if (n == 0) return G1::identity();
  std::vector<uint64_t> k(4 * n);
  for (size_t i = 0; i < n; i++) scalars[i].to_canonical(&k[4 * i]);
  const unsigned c = msm_window_bits(n);
  const unsigned nbits = 255;
  std::vector<G1> window_sums;
  std::vector<G1> buckets((size_t(1) << c) - 1);
  for (unsigned w = 0; w < nbits; w += c) {
    for (auto& b : buckets) b = G1::identity();
    G1 res = G1::identity();
    for (size_t i = 0; i < n; i++) {
      if (bases[i].inf) continue;
      // extract c bits at offset w
      unsigned limb = w / 64, off = w % 64;
      uint64_t d = k[4 * i + limb] >> off;
      if (off + c > 64 && limb + 1 < 4) d |= k[4 * i + limb + 1] << (64 - off);
      d &= (uint64_t(1) << c) - 1;
      if (d) buckets[d - 1] = g1_add_mixed(buckets[d - 1], bases[i]);
    }
    G1 running = G1::identity();
    for (size_t b = buckets.size(); b-- > 0;) {
      running = g1_add(running, buckets[b]);
      res = g1_add(res, running);
    }
    window_sums.push_back(res);
  }
  G1 total = window_sums.back();
  for (size_t w = window_sums.size() - 1; w-- > 0;) {
    for (unsigned i = 0; i < c; i++) total = g1_double(total);
    total = g1_add(total, window_sums[w]);
  }
  return total;
}
static inline G1 g1_msm(const std::vector<G1Aff>& b, const std::vector<Fr>& s) {
  return g1_msm(b.data(), s.data(), b.size());
}
// util.rs:25-29
static inline G1 g1_msm_from_projective(const std::vector<G1>& b, const std::vector<Fr>& s) {
  std::vector<G1Aff> a = g1_batch_to_affine(b);
  return g1_msm(a.data(), s.data(), a.size());
}
// Naive sum of scalar multiples — independent cross-check for the Pippenger above.
static inline G1 g1_msm_naive(const G1Aff* bases, const Fr* scalars, size_t n) {
  G1 acc = G1::identity();
  for (size_t i = 0; i < n; i++) acc = g1_add(acc, g1_mul(bases[i], scalars[i]));
  return acc;
}

}  // namespace orc
