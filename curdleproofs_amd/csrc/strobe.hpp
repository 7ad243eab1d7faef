// Keccak-f[1600] + the STROBE-128 / Merlin operations the Curdleproofs transcript performs — product code shared by
// the host engine and the device (one source for both: the GPU hashes the long instance prefix of every transcript,
// the host continues from the exported state).
// merlin 3.0.0 semantics (transcript.rs:28-60 in the reference uses merlin::Transcript): STROBE-128 over
// Keccak-f[1600], rate 166, protocol label "Merlin v1.0"; `append_message` = meta-AD(label || LE32(len)) + AD(data),
// `challenge_bytes` = meta-AD(label || LE32(len)) + PRF(len).
#pragma once
#include <cstdint>
#include <cstddef>
#include "mont32.hpp"

namespace cpx {

CPX_HD uint64_t rol64(uint64_t x, unsigned s) { return (x << s) | (x >> ((64 - s) & 63)); }

CPX_HD void keccak_f1600(uint64_t* a) {
  const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                           0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                           0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                           0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                           0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                           0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  uint64_t a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3], a04 = a[4], a05 = a[5], a06 = a[6], a07 = a[7], a08 = a[8], a09 = a[9],
           a10 = a[10], a11 = a[11], a12 = a[12], a13 = a[13], a14 = a[14], a15 = a[15], a16 = a[16], a17 = a[17], a18 = a[18],
           a19 = a[19], a20 = a[20], a21 = a[21], a22 = a[22], a23 = a[23], a24 = a[24];
  for (int r = 0; r < 24; r++) {
    // theta
    const uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21, c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22,
                   c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23, c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;
    const uint64_t d0 = c4 ^ rol64(c1, 1), d1 = c0 ^ rol64(c2, 1), d2 = c1 ^ rol64(c3, 1), d3 = c2 ^ rol64(c4, 1), d4 = c3 ^ rol64(c0, 1);
    a00 ^= d0; a05 ^= d0; a10 ^= d0; a15 ^= d0; a20 ^= d0;
    a01 ^= d1; a06 ^= d1; a11 ^= d1; a16 ^= d1; a21 ^= d1;
    a02 ^= d2; a07 ^= d2; a12 ^= d2; a17 ^= d2; a22 ^= d2;
    a03 ^= d3; a08 ^= d3; a13 ^= d3; a18 ^= d3; a23 ^= d3;
    a04 ^= d4; a09 ^= d4; a14 ^= d4; a19 ^= d4; a24 ^= d4;
    // rho + pi: B[y + 5*((2x+3y)%5)] = rot(A[x+5y], r[x][y])
    const uint64_t b00 = a00, b10 = rol64(a01, 1), b20 = rol64(a02, 62), b05 = rol64(a03, 28), b15 = rol64(a04, 27);
    const uint64_t b16 = rol64(a05, 36), b01 = rol64(a06, 44), b11 = rol64(a07, 6), b21 = rol64(a08, 55), b06 = rol64(a09, 20);
    const uint64_t b07 = rol64(a10, 3), b17 = rol64(a11, 10), b02 = rol64(a12, 43), b12 = rol64(a13, 25), b22 = rol64(a14, 39);
    const uint64_t b23 = rol64(a15, 41), b08 = rol64(a16, 45), b18 = rol64(a17, 15), b03 = rol64(a18, 21), b13 = rol64(a19, 8);
    const uint64_t b14 = rol64(a20, 18), b24 = rol64(a21, 2), b09 = rol64(a22, 61), b19 = rol64(a23, 56), b04 = rol64(a24, 14);
    // chi
    a00 = b00 ^ (~b01 & b02); a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01);
    a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06);
    a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);
    a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);
    a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);
    a00 ^= RC[r];
  }
  a[0] = a00; a[1] = a01; a[2] = a02; a[3] = a03; a[4] = a04; a[5] = a05; a[6] = a06; a[7] = a07; a[8] = a08; a[9] = a09;
  a[10] = a10; a[11] = a11; a[12] = a12; a[13] = a13; a[14] = a14; a[15] = a15; a[16] = a16; a[17] = a17; a[18] = a18; a[19] = a19;
  a[20] = a20; a[21] = a21; a[22] = a22; a[23] = a23; a[24] = a24;
}

// STROBE-128 duplex state as 25 little-endian 64-bit lanes (byte i of the state = bits 8(i%8).. of lane i/8)
struct Strobe {
  static constexpr uint32_t RATE = 166, FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_M = 16, FLAG_K = 32;
  uint64_t st[25];
  uint32_t pos, pos_begin;

  CPX_HD void xor_byte(uint32_t i, uint8_t b) { st[i >> 3] ^= (uint64_t)b << (8 * (i & 7)); }
  CPX_HD uint8_t take_byte(uint32_t i) {   // read and clear (the PRF operation overwrites the squeezed bytes with zero)
    const unsigned sh = 8 * (i & 7);
    const uint8_t b = (uint8_t)(st[i >> 3] >> sh);
    st[i >> 3] &= ~((uint64_t)0xff << sh);
    return b;
  }
  CPX_HD void run_f() {
    xor_byte(pos, (uint8_t)pos_begin);
    xor_byte(pos + 1, 0x04);
    xor_byte(RATE + 1, 0x80);
    keccak_f1600(st);
    pos = pos_begin = 0;
  }
  CPX_HD void absorb(const uint8_t* d, size_t n) {
    size_t i = 0;
    while (i < n) {
      if ((pos & 7) == 0 && n - i >= 8 && pos + 8 <= RATE) {   // a whole lane of the state at once (the instance prefix is 48 KB)
        uint64_t v = 0;
        CPX_UNROLL for (int j = 0; j < 8; j++) v |= (uint64_t)d[i + j] << (8 * j);
        st[pos >> 3] ^= v;
        pos += 8;   // (RATE is not a multiple of 8: the boundary is always reached byte by byte)
        i += 8;
        continue;
      }
      xor_byte(pos, d[i++]);
      if (++pos == RATE) run_f();
    }
  }
  CPX_HD void begin_op(uint32_t flags, bool more) {
    if (more) return;
    const uint8_t hdr[2] = {(uint8_t)pos_begin, (uint8_t)flags};
    pos_begin = pos + 1;
    absorb(hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && pos != 0) run_f();
  }
  CPX_HD void meta_ad(const void* d, size_t n, bool more) {
    begin_op(FLAG_M | FLAG_A, more);
    absorb(static_cast<const uint8_t*>(d), n);
  }
  static CPX_HD size_t label_len(const char* s) {
    size_t n = 0;
    while (s[n]) n++;
    return n;
  }

  // merlin::Transcript::new(label)
  CPX_HD void init(const char* label) {
    for (int i = 0; i < 25; i++) st[i] = 0;
    const uint8_t ini[18] = {1, 168, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    for (int i = 0; i < 18; i++) xor_byte(i, ini[i]);
    keccak_f1600(st);
    pos = pos_begin = 0;
    meta_ad("Merlin v1.0", 11, false);
    append_message("dom-sep", reinterpret_cast<const uint8_t*>(label), label_len(label));
  }
  // append_message split in two so that long messages can be streamed: begin(label, total length) then absorb(...)
  CPX_HD void append_begin(const char* label, size_t len) {
    const uint8_t l4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    meta_ad(label, label_len(label), false);
    meta_ad(l4, 4, true);
    begin_op(FLAG_A, false);
  }
  CPX_HD void append_message(const char* label, const uint8_t* m, size_t len) {
    append_begin(label, len);
    absorb(m, len);
  }
  CPX_HD void challenge_bytes(const char* label, uint8_t* out, size_t len) {
    const uint8_t l4[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    meta_ad(label, label_len(label), false);
    meta_ad(l4, 4, true);
    begin_op(FLAG_I | FLAG_A | FLAG_C, false);
    for (size_t i = 0; i < len; i++) {
      out[i] = take_byte(pos);
      if (++pos == RATE) run_f();
    }
  }
  // CurdleproofsTranscript::get_and_append_challenge (transcript.rs:40-60): 64 challenge bytes, Fr::from_random_bytes
  // on them (first 32 bytes, top bit cleared, must be canonical), retried until non-zero; the scalar is appended
  // back under the same label.  Output: canonical little-endian limbs (NOT Montgomery).
  CPX_HD void challenge_scalar_canonical(const char* label, uint32_t e[8]) {
    for (;;) {
      uint8_t buf[64];
      challenge_bytes(label, buf, 64);
      buf[31] &= 0x7f;
      bool nz = false;
      for (int i = 0; i < 8; i++) {
        e[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) | ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
        nz |= e[i] != 0;
      }
      bool lt = false;   // e < r ?
      for (int i = 7; i >= 0; i--) {
        if (e[i] != FrCfg::P[i]) {
          lt = e[i] < FrCfg::P[i];
          break;
        }
      }
      if (lt && nz) {
        append_message(label, buf, 32);   // append_scalar: the canonical 32 bytes
        return;
      }
    }
  }
};

}  // namespace cpx
