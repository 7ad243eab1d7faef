// Lane-parallel STROBE-128 / Keccak-f[1600] for the device side of the transcripts — product code (device only).
//
// Half a wave (32 lanes) per transcript: 25 of the lanes hold one 64-bit word of the Keccak state each — word x + 5 y sits in lane
// 5 y + x (+ 1 for y >= 3): lane 15 stays empty so that no plane (the five words of one y) straddles the two 16-lane DPP rows of
// the group.  A round is then ONE ds_bpermute stage (rho + pi, a gather over the 25 lanes) and DPP / permlane stages: theta's
// column parities (four row shifts by 5 / 10 lanes inside a 16-lane row + one v_permlane16_swap between the two rows), theta's
// neighbour columns (a column parity is the same in every plane, so lane l - 1 / l + 1 has it) and chi's x + 1, x + 2 inside the
// plane — instead of ~150 dependent 64-bit operations on one lane (a single GPU lane needs ~17 us per permutation; this form
// 4.8 us alone on a SIMD and 4.9 us with 4096 groups in flight; with the parities as four ds_bpermute shuffles — rounds 1-2 — 5.4
// and 7.1 us: scripts/micro/keccak_micro.hip, which also checks it against the host permutation; profiles/r03_keccak_micro.txt).
// Message bytes are absorbed by the lane that owns their position in the rate.  Same semantics as strobe.hpp (the host's code, merlin 3.0.0 as used by transcript.rs:28-60 of
// the reference), pinned by the bit-exact proof tests.  When a whole 64-lane wave drives ONE transcript, both halves run
// the same operations on the same inputs and hold identical copies of the state.
#pragma once
#include <hip/hip_runtime.h>
#include "mont32.hpp"
#include "strobe.hpp"

namespace cpx {

// bytes written to LDS by some lanes of a wave become visible to the others (no block barrier: the two proofs of a
// work-group run divergent retry loops)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct WaveStrobe {
  uint64_t a;                 // this lane's 64-bit word of the state (lanes without a word keep 0)
  uint32_t pos, pos_begin;    // uniform over the 32 lanes of a proof
  int l;                      // lane within the 32-lane group
  int w;                      // the state word this lane holds (0..24), -1 for the empty lanes 15 and 26..31

  static __device__ __forceinline__ int lane_of_word(int word) { return word + (word >= 15 ? 1 : 0); }
  static __device__ __forceinline__ int word_of_lane(int lane) { return lane < 15 ? lane : (lane >= 16 && lane <= 25 ? lane - 1 : -1); }
  __device__ __forceinline__ void set_lane(int lane_in_group) {
    l = lane_in_group;
    w = word_of_lane(l);
  }
  template <int CTRL> static __device__ __forceinline__ uint64_t dpp64(uint64_t v) {   // out-of-row sources read as zero
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, true);
    return ((uint64_t)hi << 32) | lo;
  }

  // v ^ (the same lane of the other 16-lane row of the 32-lane group), in every lane: v_permlane16_swap_b32 swaps the odd rows of
  // its first operand with the even rows of the second; with both operands = v the two results hold (row 0, row 0, row 2, row 2) and
  // (row 1, row 1, row 3, row 3)
  static __device__ __forceinline__ uint64_t rows_xor(uint64_t v) {
    const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)v, (uint32_t)v, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(v >> 32), (uint32_t)(v >> 32), false, false);
    return ((uint64_t)(hi[0] ^ hi[1]) << 32) | (uint64_t)(lo[0] ^ lo[1]);
  }
  __device__ __forceinline__ uint64_t shfl(uint64_t v, int src) const {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 32), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 32);
    return ((uint64_t)hi << 32) | lo;
  }
  __device__ void keccak() {
    const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                             0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                             0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                             0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                             0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                             0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    // rotation offsets r[x + 5y] of rho
    const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    constexpr int ROW_SHL = 0x100, ROW_SHR = 0x110;   // DPP row shifts: row_shl:n reads lane i + n, row_shr:n lane i - n (within 16 lanes)
    const int wc = w >= 0 ? w : 0;   // empty lanes run along on word 0's indices (nobody reads them, they are cleared at the end)
    const int x = wc % 5, y = wc / 5;
    // rho + pi as a gather: destination (X, Y) = (y, (2x + 3y) % 5)  <=>  source x = (X + 3Y) % 5, y = X
    const int sx = (x + 3 * y) % 5, sy = x, src = lane_of_word(sx + 5 * sy), rot = RHO[sx + 5 * sy];
    // a column parity is the same in all planes: lane l - 1 holds column x - 1 and lane l + 1 column x + 1, except next to the
    // row ends and the empty lanes, where the lane four further holds the same column
    const bool m1_alt = l == 0 || l == 16, p1_alt = l == 14 || l == 25;
    for (int r = 0; r < 24; r++) {
      // theta's column parities without LDS traffic: the planes of a row sit 5 lanes apart, so four DPP row shifts (out-of-row and
      // empty lanes read as zero) add up a column inside each 16-lane row, and ONE v_permlane16_swap (gfx950) exchanges the two
      // rows of the group: r[0] ^ r[1] = row 0's sums ^ row 1's sums in every lane.  (Until round 3 of this build: four
      // ds_bpermute shuffles — a round had two dependent LDS round trips, now one: the rho + pi gather below.)
      const uint64_t am = w >= 0 ? a : 0;   // (the empty lanes run along with garbage: keep it out of the sums)
      uint64_t c = am ^ dpp64<ROW_SHL + 5>(am) ^ dpp64<ROW_SHL + 10>(am) ^ dpp64<ROW_SHR + 5>(am) ^ dpp64<ROW_SHR + 10>(am);
      c = rows_xor(c);
      // (both candidates are fetched by every lane, then one is picked: a DPP read inside a divergent branch would see disabled lanes)
      const uint64_t cm_a = dpp64<ROW_SHL + 4>(c), cm_b = dpp64<ROW_SHR + 1>(c), cp_a = dpp64<ROW_SHR + 4>(c), cp_b = dpp64<ROW_SHL + 1>(c);
      const uint64_t cm1 = m1_alt ? cm_a : cm_b, cp1 = p1_alt ? cp_a : cp_b;
      a ^= cm1 ^ rol64(cp1, 1);
      uint64_t b = shfl(a, src);
      b = rot ? ((b << rot) | (b >> (64 - rot))) : b;
      const uint64_t b1_a = dpp64<ROW_SHL + 1>(b), b1_b = dpp64<ROW_SHR + 4>(b), b2_a = dpp64<ROW_SHL + 2>(b), b2_b = dpp64<ROW_SHR + 3>(b);
      const uint64_t b1 = x < 4 ? b1_a : b1_b, b2 = x < 3 ? b2_a : b2_b;   // (x + 1) % 5, (x + 2) % 5 of the same plane
      a = b ^ (~b1 & b2);
      if (w == 0) a ^= RC[r];
    }
    if (w < 0) a = 0;
  }
  __device__ __forceinline__ void xor_byte(uint32_t i, uint8_t b) {
    if ((int)(i >> 3) == w) a ^= (uint64_t)b << (8 * (i & 7));
  }
  __device__ void run_f() {
    xor_byte(pos, (uint8_t)pos_begin);
    xor_byte(pos + 1, 0x04);
    xor_byte(Strobe::RATE + 1, 0x80);
    keccak();
    pos = pos_begin = 0;
  }
  // absorb n bytes readable by every lane (global, constant or LDS memory)
  __device__ void absorb(const uint8_t* d, size_t n) {
    while (n) {
      const uint32_t k = (uint32_t)min((size_t)(Strobe::RATE - pos), n);
      uint64_t v = 0;
      CPX_UNROLL for (int j = 0; j < 8; j++) {
        const uint32_t sidx = 8u * (uint32_t)w + j;   // state byte owned by this lane (w = -1: beyond every position)
        if (w >= 0 && sidx >= pos && sidx < pos + k) v |= (uint64_t)d[sidx - pos] << (8 * j);
      }
      a ^= v;
      pos += k;
      d += k;
      n -= k;
      if (pos == Strobe::RATE) run_f();
    }
  }
  __device__ void begin_op(uint32_t flags, bool more) {
    if (more) return;
    const uint8_t h0 = (uint8_t)pos_begin, h1 = (uint8_t)flags;
    pos_begin = pos + 1;
    xor_byte(pos, h0);                 // the two header bytes, one at a time (the rate boundary may fall between them)
    if (++pos == Strobe::RATE) run_f();
    xor_byte(pos, h1);
    if (++pos == Strobe::RATE) run_f();
    if ((flags & (Strobe::FLAG_C | Strobe::FLAG_K)) && pos != 0) run_f();
  }
  __device__ void meta_ad(const void* d, size_t n, bool more) {
    begin_op(Strobe::FLAG_M | Strobe::FLAG_A, more);
    absorb(static_cast<const uint8_t*>(d), n);
  }
  __device__ void len4(size_t len, uint8_t* scratch4) {   // LE32(len) through the group's scratch bytes
    if (l < 4) scratch4[l] = (uint8_t)(len >> (8 * l));
    wave_lds_sync();
    meta_ad(scratch4, 4, true);
  }
  __device__ void append_begin(const char* label, size_t label_len, size_t len, uint8_t* scratch) {
    meta_ad(label, label_len, false);
    len4(len, scratch);
    begin_op(Strobe::FLAG_A, false);
  }
  __device__ void init(const char* label, size_t label_len, uint8_t* scratch) {
    const uint8_t ini[18] = {1, 168, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    a = 0;
    CPX_UNROLL for (int i = 0; i < 18; i++) xor_byte(i, ini[i]);
    keccak();
    pos = pos_begin = 0;
    meta_ad("Merlin v1.0", 11, false);
    append_begin("dom-sep", 7, label_len, scratch);
    absorb(reinterpret_cast<const uint8_t*>(label), label_len);
  }
  // ---- state import / export ([27] u64: 25 lanes, pos, pos_begin — the layout host::Transcript::load_state reads) ----
  __device__ __forceinline__ void load(const uint64_t* st27, int lane_in_group) {
    set_lane(lane_in_group);
    a = w >= 0 ? st27[w] : 0;
    pos = (uint32_t)st27[25];
    pos_begin = (uint32_t)st27[26];
  }
  __device__ __forceinline__ void store(uint64_t* st27) const {
    if (w >= 0) st27[w] = a;
    if (l == 26) st27[25] = pos;
    if (l == 27) st27[26] = pos_begin;
  }
  // append_message(label, data): data readable by every lane (global / LDS / constant)
  __device__ void append_message(const char* label, size_t label_len, const uint8_t* data, size_t len, uint8_t* scratch) {
    append_begin(label, label_len, len, scratch);
    absorb(data, len);
  }
  // CurdleproofsTranscript::append(label, &Fr) (transcript.rs:29-33): the canonical 32 little-endian bytes of a scalar every
  // lane holds (Montgomery form)
  __device__ void append_scalar(const char* label, size_t label_len, const Fr& x_mont, uint8_t* scratch) {
    const Fr c = fe_from_mont(x_mont);
    if (l < 8) {
      CPX_UNROLL for (int j = 0; j < 4; j++) scratch[16 + 4 * l + j] = (uint8_t)(c.v[l] >> (8 * j));
    }
    wave_lds_sync();
    append_begin(label, label_len, 32, scratch);
    absorb(scratch + 16, 32);
    wave_lds_sync();   // scratch may be rewritten right away by the next operation
  }
  // get_and_append_challenge (transcript.rs:40-60): 64 PRF bytes, Fr::from_random_bytes on the first 32 with the top bit
  // cleared, retried until canonical and non-zero, then appended back under the same label.  Returns the Montgomery form
  // (uniform over the lanes of the group).
  __device__ Fr challenge_scalar(const char* label, size_t label_len, uint8_t* scratch) {
    for (;;) {
      meta_ad(label, label_len, false);
      len4(64, scratch);
      begin_op(Strobe::FLAG_I | Strobe::FLAG_A | Strobe::FLAG_C, false);   // forces a permutation: pos = 0 afterwards
      // squeeze 64 bytes = lanes 0..7 of the state, which the PRF operation overwrites with zero (pos is 0 here)
      uint64_t sq[4];
      CPX_UNROLL for (int j = 0; j < 4; j++) sq[j] = shfl(a, lane_of_word(j));
      if (w >= 0 && w < 8) a = 0;
      pos = 64;
      sq[3] &= 0x7fffffffffffffffULL;
      Fr c;
      CPX_UNROLL for (int j = 0; j < 4; j++) {
        c.v[2 * j] = (uint32_t)sq[j];
        c.v[2 * j + 1] = (uint32_t)(sq[j] >> 32);
      }
      bool nz = false, lt = false;
      for (int j = 0; j < 8; j++) nz |= c.v[j] != 0;
      for (int j = 7; j >= 0; j--) {
        if (c.v[j] != FrCfg::P[j]) {
          lt = c.v[j] < FrCfg::P[j];
          break;
        }
      }
      if (lt && nz) {
        if (l < 4) {
          CPX_UNROLL for (int j = 0; j < 8; j++) scratch[16 + 8 * l + j] = (uint8_t)(sq[l] >> (8 * j));
        }
        wave_lds_sync();
        append_begin(label, label_len, 32, scratch);
        absorb(scratch + 16, 32);
        wave_lds_sync();
        return fe_to_mont(c);
      }
    }
  }
};

}  // namespace cpx
