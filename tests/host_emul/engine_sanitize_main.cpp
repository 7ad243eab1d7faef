// TEST INFRASTRUCTURE: the engine's host code (engine.cpp, engine_device.cpp, capi.cpp, whisk.cpp and the launchers of the .hip files)
// under AddressSanitizer / ThreadSanitizer ON THE GPU BOX.  scripts/sanitize_engine.sh compiles the product's sources with
// -Xarch_host -fsanitize=... (device code unchanged) together with this driver into ONE executable per sanitizer — no dlopen, no
// Python in the process — and tests/test_sanitizers.py writes the input file (instances and expected proofs from the oracle) and runs it.
//
//   engine_san <input file>
//
// Input: blobs (u64 little-endian length + bytes):  nsets | then per set:  ell | crs points | count | count x
//   ( vec_R | vec_S | vec_T | vec_U | M | permutation | k | vec_m_blinders | prover rand | verifier rand | expected proof )
// Scenarios (all through include/cpx.h, bytes and verdicts checked):
//   1. one context: a lone proof (host-driven path: spin team, worker pool), then the same instance 96 times (device-resident path: plans,
//      streams, events, fused rounds), options changed in between (plan rebuild), the fused verifier, a wrong proof;
//   2. four contexts on four threads, each walking through both CRS (set_crs: the shared registry of CRS tables) with 96-proof and
//      3-proof batches — what tests/test_gpu_parity.py::test_four_contexts_switching_crs_concurrently_stay_bit_exact does from Python;
//   3. a 128-proof batch on one context beside 1536-proof batches on another (two threads): the exclusive-SIMD kernels of the small batch
//      next to a GPU-filling one;
//   4. contexts destroyed while others still run; every buffer released at exit (LeakSanitizer).
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>
#include "cpx.h"
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#include <sanitizer/lsan_interface.h>
#define ENGINE_SAN_LEAK_CHECK() __lsan_do_leak_check()
#endif
#endif
#ifndef ENGINE_SAN_LEAK_CHECK
#define ENGINE_SAN_LEAK_CHECK() ((void)0)
#endif

namespace {

typedef std::vector<uint8_t> Bytes;
struct Inst {
  Bytes R, S, T, U, M, perm, k, mbl, prand, vrand, proof;
};
struct Set {
  size_t ell = 0;
  Bytes crs;
  std::vector<Inst> inst;
};

Bytes read_blob(FILE* f) {
  uint64_t n = 0;
  if (fread(&n, 8, 1, f) != 1) {
    fprintf(stderr, "input truncated\n");
    exit(2);
  }
  Bytes b(n);
  if (n && fread(b.data(), 1, n, f) != n) {
    fprintf(stderr, "input truncated\n");
    exit(2);
  }
  return b;
}
uint64_t as_u64(const Bytes& b) {
  uint64_t v = 0;
  memcpy(&v, b.data(), 8);
  return v;
}

std::atomic<int> g_fail{0};
#define CHECK(cond, what)                                                                               \
  do {                                                                                                  \
    if (!(cond)) {                                                                                      \
      fprintf(stderr, "engine_san FAILED (%s:%d): %s (%s)\n", __FILE__, __LINE__, what, ctx ? cpx_last_error(ctx) : ""); \
      g_fail++;                                                                                         \
      return;                                                                                           \
    }                                                                                                   \
  } while (0)

// `copies` times the instances of a set, concatenated field by field
struct Batch {
  size_t count = 0;
  Bytes R, S, T, U, M, perm, k, mbl, prand, vrand, proofs;
};
Batch make_batch(const Set& s, size_t count) {
  Batch b;
  b.count = count;
  for (size_t i = 0; i < count; i++) {
    const Inst& x = s.inst[i % s.inst.size()];
    auto app = [](Bytes& d, const Bytes& a) { d.insert(d.end(), a.begin(), a.end()); };
    app(b.R, x.R), app(b.S, x.S), app(b.T, x.T), app(b.U, x.U), app(b.M, x.M), app(b.perm, x.perm), app(b.k, x.k), app(b.mbl, x.mbl);
    app(b.prand, x.prand), app(b.vrand, x.vrand), app(b.proofs, x.proof);
  }
  return b;
}

// load, prove, compare with the expected bytes, verify, verify with one wrong proof
void prove_verify(cpx_ctx* ctx, const Set& s, const Batch& b, bool fused_too) {
  CHECK(cpx_batch_load(ctx, b.count, b.R.data(), b.S.data(), b.T.data(), b.U.data(), b.M.data()) == CPX_OK, "cpx_batch_load");
  const size_t psz = cpx_proof_size(ctx);
  CHECK(psz * b.count == b.proofs.size(), "proof size");
  Bytes out(b.proofs.size());
  CHECK(cpx_batch_prove(ctx, reinterpret_cast<const uint32_t*>(b.perm.data()), b.k.data(), b.mbl.data(), b.prand.data(), out.data()) == CPX_OK, "cpx_batch_prove");
  CHECK(out == b.proofs, "proof bytes differ from the oracle's");
  std::vector<int> verdict(b.count, CPX_ERR_INTERNAL);
  CHECK(cpx_batch_verify(ctx, out.data(), b.vrand.data(), verdict.data()) == CPX_OK, "cpx_batch_verify");
  for (int v : verdict) CHECK(v == CPX_OK, "a valid proof was rejected");
  const size_t victim = b.count / 2;
  uint8_t tmp[48];
  memcpy(tmp, &out[victim * psz], 48);
  memcpy(&out[victim * psz], &out[victim * psz + 48], 48);
  memcpy(&out[victim * psz + 48], tmp, 48);
  CHECK(cpx_batch_verify(ctx, out.data(), b.vrand.data(), verdict.data()) == CPX_OK, "cpx_batch_verify (one wrong proof)");
  for (size_t i = 0; i < b.count; i++) CHECK(verdict[i] == (i == victim ? CPX_ERR_VERIFY : CPX_OK), "verdicts with one wrong proof");
  if (fused_too) {
    Bytes frand;   // 12 factors per proof: the 8 verifier factors of the instance and 4 of its neighbour's (any non-zero reduced elements)
    for (size_t i = 0; i < b.count; i++) {
      frand.insert(frand.end(), b.vrand.begin() + i * 256, b.vrand.begin() + (i + 1) * 256);
      const size_t j = (i + 1) % b.count;
      frand.insert(frand.end(), b.vrand.begin() + j * 256, b.vrand.begin() + j * 256 + 128);
    }
    uint8_t part[144], sum[144];
    int bad = -1, is_id = -1;
    CHECK(cpx_batch_verify_fused(ctx, b.proofs.data(), frand.data(), part, &bad) == CPX_OK && bad == 0, "cpx_batch_verify_fused");
    CHECK(cpx_g1_sum_jac(ctx, part, 1, sum, &is_id) == CPX_OK && is_id == 1, "fused check of a valid batch");
    CHECK(cpx_batch_verify_fused(ctx, out.data(), frand.data(), part, &bad) == CPX_OK && bad == 0, "cpx_batch_verify_fused (wrong proof)");
    CHECK(cpx_g1_sum_jac(ctx, part, 1, sum, &is_id) == CPX_OK && is_id == 0, "fused check must refuse a batch with a wrong proof");
  }
}

void scenario_one_context(const std::vector<Set>& sets) {
  cpx_ctx* ctx = nullptr;
  CHECK(cpx_ctx_create(0, &ctx) == CPX_OK, "cpx_ctx_create");
  const Set& s = sets[0];
  CHECK(cpx_ctx_set_crs(ctx, s.ell, s.crs.data(), s.crs.size() / 96) == CPX_OK, "cpx_ctx_set_crs");
  prove_verify(ctx, s, make_batch(s, 1), true);          // host-driven
  prove_verify(ctx, s, make_batch(s, 5), false);         // host-driven with the spin team
  prove_verify(ctx, s, make_batch(s, 96), true);         // device-resident, fused rounds
  CHECK(cpx_ctx_set_option(ctx, "fused_rounds_max", 0) == CPX_OK && cpx_ctx_set_option(ctx, "fused_smsm_max", 0) == CPX_OK, "set_option");
  prove_verify(ctx, s, make_batch(s, 96), false);        // plans rebuilt: the chain of separate kernels
  CHECK(cpx_ctx_set_option(ctx, "fused_rounds_max", 256) == CPX_OK && cpx_ctx_set_option(ctx, "fused_smsm_max", 256) == CPX_OK &&
            cpx_ctx_set_option(ctx, "device_min_batch", 1) == CPX_OK,
        "set_option");
  prove_verify(ctx, s, make_batch(s, 2), false);         // device-resident on two proofs
  CHECK(cpx_set_host_threads(ctx, 3) == CPX_OK && cpx_ctx_set_option(ctx, "device_min_batch", 1000000) == CPX_OK, "set_host_threads / set_option");
  prove_verify(ctx, s, make_batch(s, 40), false);        // host-driven on 40 proofs: the worker pool
  cpx_ctx_destroy(ctx);
}

void scenario_four_contexts(const std::vector<Set>& sets) {
  std::vector<std::thread> th;
  for (int t = 0; t < 4; t++)
    th.emplace_back([&sets, t] {
      cpx_ctx* ctx = nullptr;
      CHECK(cpx_ctx_create(0, &ctx) == CPX_OK, "cpx_ctx_create");
      for (int round = 0; round < 4; round++) {
        const Set& s = sets[(t + round) % sets.size()];
        CHECK(cpx_ctx_set_crs(ctx, s.ell, s.crs.data(), s.crs.size() / 96) == CPX_OK, "cpx_ctx_set_crs");
        prove_verify(ctx, s, make_batch(s, round % 2 ? 3 : 96), round == 0);
        if (g_fail) break;
      }
      cpx_ctx_destroy(ctx);   // (the other threads' contexts go on: the CRS registry drops this reference only)
    });
  for (auto& x : th) x.join();
}

void scenario_small_beside_big(const std::vector<Set>& sets) {
  const Set& s = sets[0];
  std::atomic<bool> stop{false};
  std::thread big([&] {
    cpx_ctx* ctx = nullptr;
    CHECK(cpx_ctx_create(0, &ctx) == CPX_OK, "cpx_ctx_create");
    CHECK(cpx_ctx_set_crs(ctx, s.ell, s.crs.data(), s.crs.size() / 96) == CPX_OK, "cpx_ctx_set_crs");
    const Batch b = make_batch(s, 1536);
    for (int i = 0; i < 3 && !stop && !g_fail; i++) prove_verify(ctx, s, b, false);
    cpx_ctx_destroy(ctx);
  });
  {
    cpx_ctx* ctx = nullptr;
    CHECK(cpx_ctx_create(0, &ctx) == CPX_OK, "cpx_ctx_create");
    CHECK(cpx_ctx_set_crs(ctx, s.ell, s.crs.data(), s.crs.size() / 96) == CPX_OK, "cpx_ctx_set_crs");
    const Batch b = make_batch(s, 128);
    for (int i = 0; i < 4 && !g_fail; i++) prove_verify(ctx, s, b, i == 0);
    cpx_ctx_destroy(ctx);
  }
  stop = true;
  big.join();
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: engine_san <input file>\n");
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  std::vector<Set> sets(as_u64(read_blob(f)));
  for (Set& s : sets) {
    s.ell = as_u64(read_blob(f));
    s.crs = read_blob(f);
    s.inst.resize(as_u64(read_blob(f)));
    for (Inst& x : s.inst) {
      x.R = read_blob(f), x.S = read_blob(f), x.T = read_blob(f), x.U = read_blob(f), x.M = read_blob(f), x.perm = read_blob(f), x.k = read_blob(f);
      x.mbl = read_blob(f), x.prand = read_blob(f), x.vrand = read_blob(f), x.proof = read_blob(f);
    }
  }
  fclose(f);
  if (cpx_device_count() < 1) {
    fprintf(stderr, "engine_san: no HIP device\n");
    return 3;
  }
  scenario_one_context(sets);
  if (!g_fail) scenario_four_contexts(sets);
  if (!g_fail) scenario_small_beside_big(sets);
  if (g_fail) return 1;
  // Every context is destroyed at this point: the leak check runs NOW, and the process then leaves without the static destructors of the
  // (uninstrumented) ROCm runtime — under ROCm's ASan runtime their frees can trip its own "device runtime unloaded" CHECK during
  // __cxa_finalize (seen once in three runs; no frame of this repository on that stack), which would turn a clean run into a failed one.
  ENGINE_SAN_LEAK_CHECK();
  printf("engine_san ok: one context (host-driven, device-resident, fused and unfused rounds, option changes), four contexts switching CRS on four threads, "
         "a 128-proof batch beside 1536-proof batches — every proof bit-exact, every verdict as expected\n");
  fflush(stdout);
  fflush(stderr);
  _exit(0);
}
