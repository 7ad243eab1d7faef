// TEST-ONLY driver for sanitizer builds (ThreadSanitizer; AddressSanitizer + UndefinedBehaviorSanitizer) of the product's host-side
// headers: host_threads.hpp (sleeping pool, spin team), host_math.hpp + strobe.hpp (the host-driven transcript and Fr algebra),
// and the host/device-portable math (mont32 / fp28 / g1_28 / modinv30 / glv / recode / quad_program) through the emulation entry
// points of emul.cpp.  Recipe and clean log: tests/host_emul/sanitize.sh, profiles/r03_sanitizers.txt; run by tests/test_sanitizers.py.
//
// The threading part reproduces how the engine uses the helpers: several engine contexts (= caller threads) at once, each with its
// OWN spin team and worker pool, every context running many short parallel loops back to back (the host-driven protocol: a loop over
// the proofs between every two MSM phases), teams engaged and released around every "call".
#include "emul.cpp"
#include <cstdio>
#include <thread>

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}
static void fill(uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; i++) p[i] = (uint8_t)rnd();
}
// a reduced field element in wire form (top bits cleared: < 2^253 resp. < 2^380)
static void fill_fr(uint8_t* p, size_t n) {
  fill(p, 32 * n);
  for (size_t i = 0; i < n; i++) p[32 * i + 31] &= 0x1f;
}
static void fill_fp(uint8_t* p, size_t n) {
  fill(p, 48 * n);
  for (size_t i = 0; i < n; i++) p[48 * i + 47] &= 0x0f;
}

int main() {
  int failures = 0;
  // ---- threads: contexts side by side
  {
    std::atomic<int> bad{0};
    auto context = [&](int id) {
      WorkerPool pool(3 + id % 3);
      SpinTeam team(2 + id % 4);
      for (int call = 0; call < 40; call++) {
        team.engage();
        for (int phase = 0; phase < 12; phase++) {
          const size_t n = 1 + (size_t)((id * 7 + call * 3 + phase) % 37);
          std::vector<uint64_t> cell(n, 0);
          const std::function<void(size_t)> f = [&](size_t i) { cell[i] += i + 1; };   // disjoint items: no synchronisation needed beyond the helpers' own
          if (phase % 3 == 2) pool.run(n, f);
          else team.run(n, f);
          for (size_t i = 0; i < n; i++)
            if (cell[i] != i + 1) bad++;
        }
        team.release();
      }
    };
    std::vector<std::thread> th;
    for (int id = 0; id < 6; id++) th.emplace_back(context, id);
    for (auto& t : th) t.join();
    for (int kind = 0; kind < 3; kind++) {
      if (emul_thread_helpers(kind, 4, 9, 60, 17) != 0) bad++;
      if (emul_thread_helpers(kind, 8, 3, 100, -1) != 0) bad++;
    }
    (void)effective_host_cores();
    printf("threads: %d failures\n", bad.load());
    failures += bad.load();
  }
  // ---- transcripts on several threads (each its own object), incl. long messages and many challenges
  {
    std::atomic<int> bad{0};
    auto hash = [&](int id) {
      std::vector<uint8_t> msg(48 * 252 + 8 + id);
      for (size_t i = 0; i < msg.size(); i++) msg[i] = (uint8_t)(i * 31 + id);
      uint8_t a[64], b[64];
      emul_challenges(msg.data(), msg.size(), a);
      emul_challenges(msg.data(), msg.size(), b);
      if (memcmp(a, b, 64)) bad++;
      host::Transcript t("curdleproofs");
      t.append_point_vec_bytes("curdleproofs_step1", msg.data(), 252);
      host::SVec v = t.get_and_append_challenges("curdleproofs_vec_a", 64);
      t.append_scalar_vec("same_perm_step1", v);
      (void)t.get_and_append_challenge("same_perm_alpha");
      uint8_t mv[32];
      emul_merlin_vector(mv);
      static const uint8_t want[4] = {0xd5, 0xa2, 0x19, 0x72};   // merlin's published vector starts d5a21972...
      if (memcmp(mv, want, 4)) bad++;
    };
    std::vector<std::thread> th;
    for (int id = 0; id < 4; id++) th.emplace_back(hash, id);
    for (auto& t : th) t.join();
    printf("transcripts: %d failures\n", bad.load());
    failures += bad.load();
  }
  // ---- portable math: every emulation entry point on random reduced inputs (out-of-bounds accesses, shifts, signed overflow)
  {
    const size_t N = 24;
    std::vector<uint8_t> a(48 * N), b(48 * N), o(48 * N * 8), s(32 * N), s2(32 * N), so(66 * N + 64 * N);
    fill_fp(a.data(), N);
    fill_fp(b.data(), N);
    fill_fr(s.data(), N);
    fill_fr(s2.data(), N);
    emul_fp_mul(a.data(), b.data(), o.data(), N);
    emul_fp_addsubnegdbl(a.data(), b.data(), o.data());
    emul_fp_inv(a.data(), o.data());
    emul_fr_mul(s.data(), s2.data(), so.data(), N);
    emul_host_fr_mul(s.data(), s2.data(), so.data(), N);
    int bad = emul_host_fr_inv_check(s.data(), N);
    emul_fr_inv(s.data(), so.data());
    emul_f28_mul(a.data(), b.data(), o.data(), N);
    emul_f28_mulsub(a.data(), b.data(), b.data(), a.data(), o.data(), N, 1);
    emul_f28_lazy(a.data(), b.data(), o.data());
    emul_f28_sqrt(a.data(), o.data());
    bad += emul_modinv30_check(a.data(), N, o.data());
    emul_fr_inv_divsteps(s.data(), N, so.data());
    emul_glv_split(s.data(), N, so.data());
    uint8_t inv2[96];
    bad += emul_inv_euclid(a.data(), inv2) ? 0 : 0;
    // points: multiples of the generator through the product's own scalar multiplication
    Aff gen;
    {
      static const uint32_t GX[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u, 0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
      static const uint32_t GY[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u, 0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
      memcpy(gen.x.v, GX, 48);   // the generator in Montgomery form (ark-bls12-381 G1_GENERATOR_X / _Y limbs)
      memcpy(gen.y.v, GY, 48);
    }
    const size_t NP = 40;
    std::vector<uint8_t> pts(96 * NP), sc(32 * NP), out(144 * 8 + 96 * 8);
    fill_fr(sc.data(), NP);
    for (size_t i = 0; i < NP; i++) {
      Aff zero = Aff::identity();
      uint8_t jac[144];
      emul_smul((const uint8_t*)&zero, (const uint8_t*)&gen, sc.data() + 32 * i, jac);   // Jacobian out
      emul_to_affine(jac, pts.data() + 96 * i);
    }
    fill_fr(sc.data(), NP);
    double mags[16];
    emul_msm(pts.data(), sc.data(), NP, out.data());
    emul_f28_msm(pts.data(), sc.data(), NP, out.data(), mags);
    emul_msm_endo(pts.data(), sc.data(), NP, 0, out.data());
    emul_msm_endo(pts.data(), sc.data(), NP, 1, out.data());
    std::vector<uint8_t> signs(NP, 0);
    for (size_t i = 0; i < NP; i += 3) signs[i] = 1;
    emul_f28_xyzz_sum(pts.data(), NP, signs.data(), out.data(), mags);
    emul_f28_xyzz_full(pts.data(), NP, signs.data(), out.data(), mags);
    emul_f28_edges(pts.data(), out.data());
    bad += emul_f28_in_subgroup(pts.data()) == 1 ? 0 : 1;
    uint8_t comp[48], aff[96];
    emul_compress(pts.data(), comp);
    emul_to_affine(out.data(), aff);
    int64_t qin[64], qout[4];
    for (int i = 0; i < 64; i++) qin[i] = (int64_t)(rnd() % 1000);
    emul_quad_program(0, qin, qout);
    emul_quad_program(1, qin, qout);
    uint64_t st[25];
    for (auto& w : st) w = rnd();
    emul_keccak(st);
    printf("math: %d failures\n", bad);
    failures += bad;
  }
  printf("%s\n", failures ? "SANITIZE DRIVER: FAILURES" : "sanitize driver: all checks passed");
  return failures ? 1 : 0;
}
