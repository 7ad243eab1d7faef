// ORACLE (test infrastructure — never linked into the product path).
//
// Model of the randomness source every reference test and the README example use:
// `StdRng::seed_from_u64(0)` (rand 0.8 via ark-std ^0.4; crs.rs:63, whisk.rs:382,418,
// curdleproofs.rs:338, README.md:76) and the samplers drawn from it:
//   Fr::rand / Fp::rand        (ark-ff UniformRand: limbs drawn LSB-limb first, top limb masked,
//                               rejection if >= modulus, limbs taken AS the Montgomery form)
//   G1Projective::rand         (ark-ec: x <- Fp::rand, greatest <- bool, point-from-x, * cofactor)
//   SliceRandom::shuffle       (rand 0.8 Fisher-Yates with widening-multiply range sampling)
// None of this is under /root/reference; the semantics are pinned by the known-answer tests
// whisk.rs:381-402 and whisk.rs:416-456, which this model must reproduce byte for byte.
#pragma once
#include "field.h"
#include "g1.h"
#include <vector>

namespace orc {

struct StdRng {   // rand 0.8 StdRng = ChaCha12, 64-bit block counter, zero stream id
  uint32_t key[8];
  uint64_t counter;
  uint32_t buf[16];
  int idx;

  explicit StdRng(uint64_t seed) {
    // rand_core 0.6 SeedableRng::seed_from_u64: PCG32 expands the u64 into the 32-byte key
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    uint64_t state = seed;
    for (int i = 0; i < 8; i++) {
      state = state * MUL + INC;
      uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
      uint32_t rot = (uint32_t)(state >> 59);
      key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    counter = 0;
    idx = 16;
  }
  static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  void refill() {
    uint32_t s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) s[4 + i] = key[i];
    s[12] = (uint32_t)counter;
    s[13] = (uint32_t)(counter >> 32);
    s[14] = s[15] = 0;
    uint32_t x[16];
    memcpy(x, s, sizeof x);
#define ORC_QR(a, b, c, d)                                  \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);               \
  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);               \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);                \
  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    for (int r = 0; r < 6; r++) {   // 12 rounds
      ORC_QR(0, 4, 8, 12) ORC_QR(1, 5, 9, 13) ORC_QR(2, 6, 10, 14) ORC_QR(3, 7, 11, 15)
      ORC_QR(0, 5, 10, 15) ORC_QR(1, 6, 11, 12) ORC_QR(2, 7, 8, 13) ORC_QR(3, 4, 9, 14)
    }
#undef ORC_QR
    for (int i = 0; i < 16; i++) buf[i] = x[i] + s[i];
    counter++;
    idx = 0;
  }
  uint32_t next_u32() {
    if (idx >= 16) refill();
    return buf[idx++];
  }
  // rand_core BlockRng::next_u64 reads two consecutive words of the continuous stream
  uint64_t next_u64() {
    uint64_t lo = next_u32();
    uint64_t hi = next_u32();
    return lo | (hi << 32);
  }
  bool next_bool() { return (int32_t)next_u32() < 0; }   // rand `Standard` for bool
};

template <class F> static inline F rand_field(StdRng& rng, int shave_bits) {
  for (;;) {
    F t;
    for (int i = 0; i < F::N; i++) t.v[i] = rng.next_u64();
    t.v[F::N - 1] &= ~uint64_t(0) >> shave_bits;
    if (big_cmp<F::N>(t.v, F::Params::MOD) < 0) return t;
  }
}
static inline Fr rand_fr(StdRng& rng) { return rand_field<Fr>(rng, 1); }   // 256 - 255
static inline Fp rand_fp(StdRng& rng) { return rand_field<Fp>(rng, 3); }   // 384 - 381

// ark-ec `Projective::rand` for bls12-381 G1: multiply by the curve cofactor h
static inline G1 rand_g1(StdRng& rng) {
  static const uint64_t COFACTOR[2] = {0x8c00aaab0000aaabULL, 0x396c8c005555e156ULL};
  for (;;) {
    Fp x = rand_fp(rng);
    bool greatest = rng.next_bool();
    G1Aff p;
    if (g1_point_from_x(x, greatest, &p)) return g1_mul_limbs(G1::from_affine(p), COFACTOR, 2);
  }
}

// rand 0.8 UniformInt<u32>::sample_single(0, range)
static inline uint32_t gen_range_u32(StdRng& rng, uint32_t range) {
  uint32_t zone = (range << __builtin_clz(range)) - 1;
  for (;;) {
    uint32_t v = rng.next_u32();
    uint64_t m = (uint64_t)v * range;
    uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
    if (lo <= zone) return hi;
  }
}
// rand 0.8 SliceRandom::shuffle
template <class T> static inline void shuffle(StdRng& rng, std::vector<T>& v) {
  for (size_t i = v.size(); i-- > 1;) {
    size_t j = gen_range_u32(rng, (uint32_t)(i + 1));
    std::swap(v[i], v[j]);
  }
}

}  // namespace orc
