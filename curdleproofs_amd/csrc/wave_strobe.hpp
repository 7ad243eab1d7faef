// Lane-parallel STROBE-128 / Keccak-f[1600] for the device side of the transcripts — product code (device only).
//
// ONE 64-lane wave per transcript, state bit-interleaved over the two halves of the wave: the 25 lanes 0..24 (+ the gaps below) of
// the lower half hold the EVEN bits of one 64-bit state word each, the same lanes of the upper half (lane + 32) its ODD bits.  In
// that form a 64-bit rotation is two independent 32-bit rotations (by r/2 on both halves for an even r; for an odd r the halves
// change places and rotate by (r+1)/2 and (r-1)/2), every bitwise step is one 32-bit operation per lane, and nothing is computed
// twice: 27 vector instructions per round against the 75 of the form this file held until round 4 (32 lanes per transcript, a
// 64-bit word = two registers per lane, both halves of a wave running identical copies: 4.8 us per permutation; this form: 2.4 us,
// profiles/r04_keccak_micro.txt).
//
// Word x + 5 y sits in lane 5 y + x (+ 1 for y >= 3) of its half: lane 15 stays empty so that no plane (the five words of one y)
// straddles the two 16-lane DPP rows of a half.  A round:
//   * theta's column parities: four DPP row shifts by 5 / 10 lanes inside a 16-lane row + one v_permlane16_swap between the two
//     rows of a half; every lane then holds the parity of ITS column (its half of the bits);
//   * theta's D, rho and pi in ONE gather stage of three ds_bpermute: rol(A ^ D, r) = rol(A, r) ^ rol(C[x-1], r) ^ rol(C[x+1], r+1)
//     is linear, so the destination lane fetches the source word and the two neighbour-column parities from the half the rotation
//     amount's parity selects and rotates them itself (per-lane constants) — the fetch of A is issued before the parities exist;
//   * chi's x + 1, x + 2 of the same plane by DPP row shifts (two candidates each, the plane's wrap-around picks one);
//   * iota: the round constant's even / odd bits into word 0 of each half.
// Message bytes are absorbed four per lane (the lower half owns bytes 0..3 of a word, the upper half bytes 4..7), un-shuffled into
// even | odd bits and exchanged with the partner lane.  Same semantics as strobe.hpp (the host's code, merlin 3.0.0 as used by
// transcript.rs:28-60 of the reference), pinned by the bit-exact proof tests and by scripts/micro/keccak_micro.hip against the host
// permutation.  Values the callers use (challenges, pos) are uniform over the wave.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "mont32.hpp"
#include "strobe.hpp"

namespace cpx {

// bytes written to LDS by some lanes of a wave become visible to the others (no block barrier needed: one wave per work-group
// or per transcript)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// perfect outer un-shuffle of 32 bits: even bits -> low 16, odd bits -> high 16 (Hacker's Delight 7-2), and its inverse
CPX_HD uint32_t bits_unshuffle32(uint32_t x) {
  uint32_t t;
  t = (x ^ (x >> 1)) & 0x22222222u; x ^= t ^ (t << 1);
  t = (x ^ (x >> 2)) & 0x0c0c0c0cu; x ^= t ^ (t << 2);
  t = (x ^ (x >> 4)) & 0x00f000f0u; x ^= t ^ (t << 4);
  t = (x ^ (x >> 8)) & 0x0000ff00u; x ^= t ^ (t << 8);
  return x;
}
CPX_HD uint32_t bits_shuffle32(uint32_t x) {
  uint32_t t;
  t = (x ^ (x >> 8)) & 0x0000ff00u; x ^= t ^ (t << 8);
  t = (x ^ (x >> 4)) & 0x00f000f0u; x ^= t ^ (t << 4);
  t = (x ^ (x >> 2)) & 0x0c0c0c0cu; x ^= t ^ (t << 2);
  t = (x ^ (x >> 1)) & 0x22222222u; x ^= t ^ (t << 1);
  return x;
}
// 64-bit word <-> (even bits, odd bits)
CPX_HD void bits_split64(uint64_t v, uint32_t& even, uint32_t& odd) {
  const uint32_t lo = bits_unshuffle32((uint32_t)v), hi = bits_unshuffle32((uint32_t)(v >> 32));
  even = (lo & 0xffffu) | (hi << 16);
  odd = (lo >> 16) | (hi & 0xffff0000u);
}
CPX_HD uint64_t bits_join64(uint32_t even, uint32_t odd) {
  const uint32_t lo = bits_shuffle32((even & 0xffffu) | (odd << 16)), hi = bits_shuffle32((even >> 16) | (odd & 0xffff0000u));
  return ((uint64_t)hi << 32) | lo;
}

struct KeccakRC {   // the 24 round constants as (even bits, odd bits)
  uint32_t e[24], o[24];
};
constexpr KeccakRC make_keccak_rc() {
  const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                           0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                           0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                           0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                           0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                           0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  KeccakRC t{};
  for (int r = 0; r < 24; r++) {
    uint32_t e = 0, o = 0;
    for (int i = 0; i < 32; i++) {
      e |= (uint32_t)((RC[r] >> (2 * i)) & 1) << i;
      o |= (uint32_t)((RC[r] >> (2 * i + 1)) & 1) << i;
    }
    t.e[r] = e;
    t.o[r] = o;
  }
  return t;
}

// Keccak-f[1600] on the wave's bit-interleaved state (layout and round structure: the header of this file).  `a`: this lane's half word
// (empty lanes zero), lane = 0..63.  ONE out-of-line copy per code object, all 24 rounds unrolled with their constants as literals: a
// kernel that hashes in a dozen places stays a few KB instead of 50-120 KB, and a round loads nothing.
__device__ __noinline__ static uint32_t wave_keccak_f1600(uint32_t a, uint32_t lane) {
  constexpr KeccakRC RC = make_keccak_rc();
  // rotation offsets r[x + 5y] of rho
  const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  constexpr int ROW_SHL = 0x100, ROW_SHR = 0x110;   // DPP row shifts: row_shl:n reads lane i + n, row_shr:n lane i - n (within 16 lanes)
  auto lane_of_word = [](int word) { return word + (word >= 15 ? 1 : 0); };
  auto dpp = [](uint32_t v, auto ctrl) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, 0xf, 0xf, true); };   // out-of-row sources read as zero
  const int l = (int)(lane & 31u), h = (int)((lane >> 5) & 1u);
  const int w = l < 15 ? l : (l >= 16 && l <= 25 ? l - 1 : -1);
  const bool live = w >= 0;
  const int wc = live ? w : 0;   // empty lanes run along on word 0's indices (nobody reads them, they are cleared every round)
  const int x = wc % 5, y = wc / 5;
  // rho + pi as a gather: destination (X, Y) = (y, (2x + 3y) % 5)  <=>  source x = (X + 3Y) % 5, y = X
  const int sx = (x + 3 * y) % 5, sy = x, rot = RHO[sx + 5 * sy];
  // where a destination lane of half hh finds its bits for a 64-bit left rotation by r (0..63): source half and 32-bit amount
  auto rot_plan = [](int r, int hh, int& src_half, int& amount) {
    if ((r & 1) == 0) {
      src_half = hh;
      amount = r >> 1;
    } else {
      src_half = hh ^ 1;
      amount = hh == 0 ? (r + 1) >> 1 : (r - 1) >> 1;
    }
  };
  int half_a, amt_a, half_p, amt_p;
  rot_plan(rot, h, half_a, amt_a);
  rot_plan(rot + 1, h, half_p, amt_p);
  // a column parity is the same in every plane of its half: take the neighbour columns from the source word's own plane
  const int addr_a = 4 * (32 * half_a + lane_of_word(sx + 5 * sy));
  const int addr_m = 4 * (32 * half_a + lane_of_word((sx + 4) % 5 + 5 * sy));
  const int addr_p = 4 * (32 * half_p + lane_of_word((sx + 1) % 5 + 5 * sy));
  const uint32_t sh_a = (uint32_t)(32 - amt_a) & 31u, sh_p = (uint32_t)(32 - amt_p) & 31u;   // v_alignbit amounts: (32 - left amount) & 31
  const bool x_lt4 = x < 4, x_lt3 = x < 3;
  // per-lane masks kept as opaque register values (the compiler would otherwise turn them back into selects on scalar masks, one
  // more move each): word 0's lane of each half for iota, the lanes that hold a word
  uint32_t iota_e = (w == 0 && h == 0) ? ~0u : 0u, iota_o = (w == 0 && h == 1) ? ~0u : 0u, live_m = live ? ~0u : 0u;
  asm volatile("" : "+v"(iota_e), "+v"(iota_o), "+v"(live_m));
  using std::integral_constant;
  CPX_UNROLL for (int r = 0; r < 24; r++) {
    const uint32_t ga = (uint32_t)__builtin_amdgcn_ds_bpermute(addr_a, (int)a);
    uint32_t c = dpp(a, integral_constant<int, ROW_SHL + 5>{}) ^ dpp(a, integral_constant<int, ROW_SHL + 10>{}) ^ dpp(a, integral_constant<int, ROW_SHR + 5>{}) ^
                 dpp(a, integral_constant<int, ROW_SHR + 10>{}) ^ a;
    {   // c ^= the same lane of the other 16-lane row of the half: v_permlane16_swap_b32 swaps the odd rows of its first operand with the
        // even rows of the second; with both = c the two results hold (row 0, row 0, row 2, row 2) and (row 1, row 1, row 3, row 3)
      const auto sw = __builtin_amdgcn_permlane16_swap(c, c, false, false);
      c = sw[0] ^ sw[1];
    }
    const uint32_t gm = (uint32_t)__builtin_amdgcn_ds_bpermute(addr_m, (int)c), gp = (uint32_t)__builtin_amdgcn_ds_bpermute(addr_p, (int)c);
    const uint32_t t = ga ^ gm;
    const uint32_t b = __builtin_amdgcn_alignbit(t, t, sh_a) ^ __builtin_amdgcn_alignbit(gp, gp, sh_p);
    // (both candidates are fetched by every lane, then one is picked: a DPP read inside a divergent branch would see disabled lanes)
    const uint32_t b1_a = dpp(b, integral_constant<int, ROW_SHL + 1>{}), b1_b = dpp(b, integral_constant<int, ROW_SHR + 4>{});
    const uint32_t b2_a = dpp(b, integral_constant<int, ROW_SHL + 2>{}), b2_b = dpp(b, integral_constant<int, ROW_SHR + 3>{});
    const uint32_t b1 = x_lt4 ? b1_a : b1_b, b2 = x_lt3 ? b2_a : b2_b;   // (x + 1) % 5, (x + 2) % 5 of the same plane
    const uint32_t rc = (RC.e[r] & iota_e) | (RC.o[r] & iota_o);
    a = (b ^ (~b1 & b2) ^ rc) & live_m;   // the empty lanes are read by the parity shifts: keep them zero
  }
  return a;
}

struct WaveStrobe {
  uint32_t a;                 // this lane's half of a state word: its even bits (lanes 0..31) or odd bits (lanes 32..63); empty lanes keep 0
  uint32_t pos, pos_begin;    // uniform over the wave
  int l;                      // lane within its 32-lane half
  int h;                      // 0: even bits, 1: odd bits
  int w;                      // the state word this lane holds (0..24), -1 for the empty lanes 15 and 26..31

  static __device__ __forceinline__ int lane_of_word(int word) { return word + (word >= 15 ? 1 : 0); }
  static __device__ __forceinline__ int word_of_lane(int lane) { return lane < 15 ? lane : (lane >= 16 && lane <= 25 ? lane - 1 : -1); }
  __device__ __forceinline__ void set_lane(int lane_in_wave) {   // 0..63
    l = lane_in_wave & 31;
    h = (lane_in_wave >> 5) & 1;
    w = word_of_lane(l);
  }
  template <int CTRL> static __device__ __forceinline__ uint32_t dpp(uint32_t v) {   // out-of-row sources read as zero
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
  }
  // v ^ (the same lane of the other 16-lane row of the half), in every lane: v_permlane16_swap_b32 swaps the odd rows of its first
  // operand with the even rows of the second; with both operands = v the two results hold (row 0, row 0, row 2, row 2) and
  // (row 1, row 1, row 3, row 3)
  static __device__ __forceinline__ uint32_t rows_xor(uint32_t v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return r[0] ^ r[1];
  }
  static __device__ __forceinline__ uint32_t gather(int byte_addr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)v); }
  static __device__ __forceinline__ uint32_t rotr_by(uint32_t v, uint32_t sh) { return __builtin_amdgcn_alignbit(v, v, sh); }   // sh = (32 - left amount) & 31
  // where a destination lane of half `hh` finds its bits for a 64-bit left rotation by r (0..63): source half and 32-bit amount
  static __device__ __forceinline__ void rot_plan(int r, int hh, int& src_half, int& amount) {
    if ((r & 1) == 0) {
      src_half = hh;
      amount = r >> 1;
    } else {
      src_half = hh ^ 1;
      amount = hh == 0 ? (r + 1) >> 1 : (r - 1) >> 1;
    }
  }
  __device__ __forceinline__ void keccak() { a = wave_keccak_f1600(a, (uint32_t)(32 * h + l)); }
  // state byte i ^= b (i, b uniform): the even (lower half) or odd (upper half) bits of the byte, four of them, at bit 4 (i % 8)
  __device__ __forceinline__ void xor_byte(uint32_t i, uint8_t b) {
    if ((int)(i >> 3) == w) {
      const uint32_t v = (uint32_t)b >> h;
      a ^= ((v & 1u) | ((v >> 1) & 2u) | ((v >> 2) & 4u) | ((v >> 3) & 8u)) << (4 * (i & 7));
    }
  }
  __device__ void run_f() {
    xor_byte(pos, (uint8_t)pos_begin);
    xor_byte(pos + 1, 0x04);
    xor_byte(Strobe::RATE + 1, 0x80);
    keccak();
    pos = pos_begin = 0;
  }
  // the lane's word of message bits `v` (bytes 0..3 of its state word in the lower half, bytes 4..7 in the upper half, zero where
  // nothing is absorbed) into the state: un-shuffled into even | odd bits, 16 bits traded with the partner lane
  __device__ __forceinline__ void xor_own_bytes(uint32_t v) {
    const uint32_t mine = bits_unshuffle32(v), other = (uint32_t)__shfl_xor((int)mine, 32, 64);
    // lower half: even bits of bytes 0..3 (mine) | even bits of bytes 4..7 (other) << 16; upper half: the odd bits likewise
    a ^= h ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
  }
  // absorb the n bytes at d (readable by every lane: global, constant or LDS memory) followed by the tail_n (<= 4) low bytes of `tail`,
  // little-endian — a label and the LE32 length that follows it are one message (merlin's meta-AD with `more`), absorbed in one pass
  // without staging the length in memory.  A lane fetches the four bytes of its word that its half owns.
  __device__ void absorb(const uint8_t* d, size_t n, uint32_t tail = 0, uint32_t tail_n = 0) {
    size_t left = n + tail_n, off = 0;
    // the lane's bytes of the chunk that starts at message offset `o` and fills the rate from position `p` with `kk` bytes
    auto fetch = [&](size_t o, uint32_t p, uint32_t kk) {
      uint32_t v = 0;
      CPX_UNROLL for (int j = 0; j < 4; j++) {
        const uint32_t sidx = 8u * (uint32_t)w + 4u * (uint32_t)h + j;   // state byte owned by this lane (w = -1: beyond every position)
        if (w >= 0 && sidx >= p && sidx < p + kk) {
          const size_t m = o + (sidx - p);
          const uint32_t b = m < n ? (uint32_t)d[m] : (tail >> (8 * (uint32_t)(m - n))) & 0xffu;
          v |= b << (8 * j);
        }
      }
      return v;
    };
    uint32_t k = (uint32_t)min((size_t)(Strobe::RATE - pos), left);
    uint32_t v = fetch(0, pos, k);
    while (left) {
      // the next chunk starts at position 0 after a permutation: its bytes are fetched BEFORE the permutation runs (a long
      // message comes from HBM, one block per permutation: the load's latency hides behind the 2.4 us of the permutation)
      const size_t left_next = left - k;
      const uint32_t k_next = (uint32_t)min((size_t)Strobe::RATE, left_next);
      const bool full = pos + k == Strobe::RATE;
      uint32_t v_next = 0;
      if (full && left_next) v_next = fetch(off + k, 0, k_next);
      xor_own_bytes(v);
      pos += k;
      off += k;
      left = left_next;
      if (full) run_f();
      v = v_next;
      k = k_next;
    }
  }
  // absorb the 32 bytes of eight words every lane holds (uniform): a lane picks the two words its four bytes fall into — no memory
  __device__ void absorb_words8(const uint32_t* c) {
    uint32_t left = 32, off = 0;
    while (left) {
      const uint32_t k = min(Strobe::RATE - pos, left);
      const int s0 = 8 * w + 4 * h;                        // this lane's first state byte
      const int o4 = s0 - (int)pos + (int)off + 4;         // its message offset + 4 (>= 1 where the lane absorbs anything)
      const int i4 = o4 >> 2;                              // words i4 - 1 (low) and i4 (high) of c hold the window (zero outside 0..7)
      uint32_t lo = 0, hi = 0;
      CPX_UNROLL for (int t = 0; t < 8; t++) {
        lo = i4 == t + 1 ? c[t] : lo;
        hi = i4 == t ? c[t] : hi;
      }
      uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)o4 & 3u);
      // keep the bytes j with pos <= s0 + j < pos + k
      const int b0 = max((int)pos - s0, 0), b1 = min((int)(pos + k) - s0, 4);
      const uint32_t m = (w >= 0 && b0 < b1) ? (uint32_t)((((uint64_t)1 << (8 * (b1 - b0))) - 1) << (8 * b0)) : 0u;
      xor_own_bytes(v & m);
      pos += k;
      off += k;
      left -= k;
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
  // meta-AD(label || LE32(len)), then the operation header of the data that follows (merlin's append_message up to the data)
  __device__ void append_begin(const char* label, size_t label_len, size_t len, uint8_t* /*scratch*/) {
    begin_op(Strobe::FLAG_M | Strobe::FLAG_A, false);
    absorb(reinterpret_cast<const uint8_t*>(label), label_len, (uint32_t)len, 4);
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
  // ---- state import / export ([27] u64: 25 plain 64-bit words, pos, pos_begin — the layout host::Transcript::load_state reads) ----
  __device__ __forceinline__ void load(const uint64_t* st27, int lane_in_wave) {
    set_lane(lane_in_wave);
    uint32_t e = 0, o = 0;
    if (w >= 0) bits_split64(st27[w], e, o);
    a = h ? o : e;
    pos = (uint32_t)st27[25];
    pos_begin = (uint32_t)st27[26];
  }
  __device__ __forceinline__ void store(uint64_t* st27) const {   // called by the whole wave
    const uint32_t other = (uint32_t)__shfl_xor((int)a, 32, 64);
    if (h == 0) {
      if (w >= 0) st27[w] = bits_join64(a, other);
      if (l == 26) st27[25] = pos;
      if (l == 27) st27[26] = pos_begin;
    }
  }
  // the first 32 bytes of the state (words 0..3) as plain 64-bit words, uniform (scalar reads of the eight lanes that hold them)
  __device__ __forceinline__ void peek4(uint64_t sq[4]) const {
    CPX_UNROLL for (int j = 0; j < 4; j++) {
      const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)a, lane_of_word(j)), o = (uint32_t)__builtin_amdgcn_readlane((int)a, 32 + lane_of_word(j));
      sq[j] = bits_join64(e, o);
    }
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
    append_begin(label, label_len, 32, scratch);
    absorb_words8(c.v);
  }
  // One attempt of get_and_append_challenge (transcript.rs:40-60): 64 PRF bytes, Fr::from_random_bytes on the first 32 with the top
  // bit cleared; when canonical and non-zero the scalar is appended back under the same label and returned in canonical form
  // (uniform over the wave), otherwise the caller retries.
  __device__ bool challenge_attempt(const char* label, size_t label_len, uint8_t* scratch, Fr& c) {
    begin_op(Strobe::FLAG_M | Strobe::FLAG_A, false);
    absorb(reinterpret_cast<const uint8_t*>(label), label_len, 64, 4);
    begin_op(Strobe::FLAG_I | Strobe::FLAG_A | Strobe::FLAG_C, false);   // forces a permutation: pos = 0 afterwards
    // squeeze 64 bytes = words 0..7 of the state, which the PRF operation overwrites with zero (pos is 0 here)
    uint64_t sq[4];
    peek4(sq);
    if (w >= 0 && w < 8) a = 0;
    pos = 64;
    sq[3] &= 0x7fffffffffffffffULL;
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
    if (!(lt && nz)) return false;
    append_begin(label, label_len, 32, scratch);
    absorb_words8(c.v);
    return true;
  }
  // get_and_append_challenge: retried until canonical and non-zero.  Returns the Montgomery form (uniform over the wave).
  __device__ Fr challenge_scalar(const char* label, size_t label_len, uint8_t* scratch) {
    Fr c;
    while (!challenge_attempt(label, label_len, scratch, c)) {
    }
    return fe_to_mont(c);
  }
};

}  // namespace cpx
