// ORACLE (test infrastructure — never linked into the product path).
//
// Restatement of the Fiat-Shamir transcript the reference builds on
// merlin 3.0.0 (Cargo.toml:21; not vendored): Keccak-f[1600] -> STROBE-128 -> Merlin,
// plus the reference's own wrapper trait `CurdleproofsTranscript`
// (/root/reference/src/transcript.rs:14-60).
// Pinned by merlin's published test vector (tests/test_oracle_kat.py::test_merlin_vector)
// and, end to end, by the reference KATs whisk.rs:381-456.
#pragma once
#include "field.h"
#include "g1.h"
#include <vector>
#include <string>

namespace orc {

static inline uint64_t rotl64(uint64_t x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }

static inline void keccak_f1600(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
      0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (int round = 0; round < 24; round++) {
    uint64_t bc[5];
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      uint64_t t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    uint64_t t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = PIL[i];
      uint64_t b = st[j];
      st[j] = rotl64(t, ROT[i]);
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= RC[round];
  }
}

// merlin::strobe::Strobe128 (only the operations Merlin uses: meta-AD, AD, PRF)
struct Strobe128 {
  static constexpr uint8_t R = 166;
  static constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_M = 16, FLAG_K = 32;
  uint8_t state[200];
  uint8_t pos, pos_begin, cur_flags;

  explicit Strobe128(const char* protocol_label) {
    memset(state, 0, sizeof state);
    const uint8_t hdr[6] = {1, (uint8_t)(R + 2), 1, 0, 1, 96};
    memcpy(state, hdr, 6);
    memcpy(state + 6, "STROBEv1.0.2", 12);
    permute();
    pos = pos_begin = cur_flags = 0;
    meta_ad((const uint8_t*)protocol_label, strlen(protocol_label), false);
  }
  void permute() {
    uint64_t l[25];
    memcpy(l, state, 200);   // little-endian host
    keccak_f1600(l);
    memcpy(state, l, 200);
  }
  void run_f() {
    state[pos] ^= pos_begin;
    state[pos + 1] ^= 0x04;
    state[R + 1] ^= 0x80;
    permute();
    pos = 0;
    pos_begin = 0;
  }
  void absorb(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
      state[pos] ^= d[i];
      if (++pos == R) run_f();
    }
  }
  void squeeze(uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) {
      d[i] = state[pos];
      state[pos] = 0;
      if (++pos == R) run_f();
    }
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;   // continuing: flags must equal cur_flags (callers below guarantee it)
    uint8_t old_begin = pos_begin;
    pos_begin = pos + 1;
    cur_flags = flags;
    uint8_t hdr[2] = {old_begin, flags};
    absorb(hdr, 2);
    bool force_f = flags & (FLAG_C | FLAG_K);
    if (force_f && pos != 0) run_f();
  }
  void meta_ad(const uint8_t* d, size_t n, bool more) {
    begin_op(FLAG_M | FLAG_A, more);
    absorb(d, n);
  }
  void ad(const uint8_t* d, size_t n, bool more) {
    begin_op(FLAG_A, more);
    absorb(d, n);
  }
  void prf(uint8_t* d, size_t n, bool more) {
    begin_op(FLAG_I | FLAG_A | FLAG_C, more);
    squeeze(d, n);
  }
};

// merlin::Transcript + the reference's CurdleproofsTranscript wrapper (transcript.rs:28-60)
struct Transcript {
  Strobe128 strobe;
  explicit Transcript(const char* label) : strobe("Merlin v1.0") { append_message("dom-sep", (const uint8_t*)label, strlen(label)); }

  void append_message(const char* label, const uint8_t* msg, size_t len) {
    uint8_t l4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    strobe.meta_ad((const uint8_t*)label, strlen(label), false);
    strobe.meta_ad(l4, 4, true);
    strobe.ad(msg, len, false);
  }
  void challenge_bytes(const char* label, uint8_t* dest, size_t len) {
    uint8_t l4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    strobe.meta_ad((const uint8_t*)label, strlen(label), false);
    strobe.meta_ad(l4, 4, true);
    strobe.prf(dest, len, false);
  }

  // transcript.rs:29-33 `append` for the item kinds the protocol hashes
  void append_fr(const char* label, const Fr& x) {
    uint8_t b[32];
    fr_to_bytes(x, b);
    append_message(label, b, 32);
  }
  void append_g1(const char* label, const G1Aff& p) {
    uint8_t b[48];
    g1_compress(p, b);
    append_message(label, b, 48);
  }
  void append_g1(const char* label, const G1& p) { append_g1(label, g1_to_affine(p)); }
  // Vec<T> serialises as u64-LE length || items (ark-serialize), one message per vector
  void append_g1_vec(const char* label, const std::vector<G1Aff>& v) {
    std::vector<uint8_t> b(8 + 48 * v.size());
    uint64_t n = v.size();
    memcpy(b.data(), &n, 8);
    for (size_t i = 0; i < v.size(); i++) g1_compress(v[i], &b[8 + 48 * i]);
    append_message(label, b.data(), b.size());
  }
  void append_fr_vec(const char* label, const std::vector<Fr>& v) {
    std::vector<uint8_t> b(8 + 32 * v.size());
    uint64_t n = v.size();
    memcpy(b.data(), &n, 8);
    for (size_t i = 0; i < v.size(); i++) fr_to_bytes(v[i], &b[8 + 32 * i]);
    append_message(label, b.data(), b.size());
  }

  // transcript.rs:41-54: 64 challenge bytes -> Fr::from_random_bytes (first 32 bytes LE, bit 255
  // cleared, None if >= r) ; retry on None or zero; feed the accepted challenge back.
  Fr get_and_append_challenge(const char* label) {
    for (;;) {
      uint8_t buf[64];
      challenge_bytes(label, buf, 64);
      buf[31] &= 0x7f;
      Fr e;
      if (fr_from_bytes(buf, &e) && !e.is_zero()) {
        append_fr(label, e);
        return e;
      }
    }
  }
  std::vector<Fr> get_and_append_challenges(const char* label, size_t n) {
    std::vector<Fr> v(n);
    for (size_t i = 0; i < n; i++) v[i] = get_and_append_challenge(label);
    return v;
  }
};

}  // namespace orc
