// extern "C" boundary (include/cpx.h) over the engine.  Nothing unwinds across it.
#include "../../include/cpx.h"
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>
#include <cstdlib>
#include "engine.hpp"

using namespace cpx;

struct cpx_ctx {
  Engine* eng = nullptr;
  std::string err;
};

// msm_accumulator.rs:22-68.  The per-base scalar merge (Fr additions keyed by the base's bytes) is host
// bookkeeping; the group work (a*C and the final MSM) runs on the device.
struct cpx_accum {
  cpx_ctx* ctx;
  struct Key {
    uint8_t b[96];
    bool operator<(const Key& o) const { return memcmp(b, o.b, 96) < 0; }
  };
  std::map<Key, host::S> base_scalar;
  std::vector<uint8_t> lhs_points;   // Jacobian C_i (144 B each)
  std::vector<host::S> lhs_scalars;  // a_i
};

namespace {
// The host side of this library is compiled for x86-64-v3 + ADX (build.py: the Keccak permutation of the host transcripts and the 64-bit-limb Fr
// products use BMI2 / ADX forms).  A host without them would die of SIGILL somewhere inside a prove; cpx_ctx_create refuses it instead.
const char* missing_cpu_feature() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)   // (this file is compiled as HIP: the device pass has no CPU to ask)
  __builtin_cpu_init();
  if (!__builtin_cpu_supports("avx2")) return "AVX2";
  if (!__builtin_cpu_supports("bmi2")) return "BMI2";
  if (!__builtin_cpu_supports("fma")) return "FMA";
  unsigned a = 0, b = 0, c = 0, d = 0;
  __asm__ volatile("cpuid" : "=a"(a), "=b"(b), "=c"(c), "=d"(d) : "a"(7u), "c"(0u));
  if (!(b & (1u << 19))) return "ADX";   // CPUID.(EAX=7, ECX=0):EBX bit 19
#endif
  return nullptr;
}
thread_local std::string g_create_error;   // why the calling thread's last cpx_ctx_create failed (there is no context to ask)
}  // namespace

namespace {
template <class F> int guarded(cpx_ctx* ctx, F&& f) {
  if (!ctx || !ctx->eng) return CPX_ERR_ARG;
  try {
    return f();
  } catch (const HipError& e) {
    ctx->err = e.what();
    return CPX_ERR_HIP;
  } catch (const ArgError& e) {
    ctx->err = e.what();
    return CPX_ERR_ARG;
  } catch (const std::invalid_argument& e) {
    ctx->err = e.what();
    return CPX_ERR_NOT_POW2;
  } catch (const std::logic_error& e) {
    ctx->err = e.what();
    return CPX_ERR_STATE;
  } catch (const std::bad_alloc&) {
    ctx->err = "out of host memory";
    return CPX_ERR_INTERNAL;
  } catch (const std::exception& e) {
    ctx->err = e.what();
    return CPX_ERR_INTERNAL;
  } catch (...) {
    ctx->err = "unknown failure";
    return CPX_ERR_INTERNAL;
  }
}
}  // namespace

extern "C" {

int cpx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void* cpx_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  CPX_HOST_MEMORY_ACQUIRE();
  return p;
}
void cpx_host_free(void* p) {
  CPX_HOST_MEMORY_RELEASE();
  if (p) (void)hipHostFree(p);
}

int cpx_ctx_create(int device_id, cpx_ctx** out) {
  if (!out) return CPX_ERR_ARG;
  *out = nullptr;
  g_create_error.clear();
  if (const char* f = missing_cpu_feature()) {
    g_create_error = std::string("cpx_ctx_create: this host CPU lacks ") + f + " (libcpx.so's host code is built for x86-64-v3 + ADX)";
    return CPX_ERR_STATE;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) {
    g_create_error = "cpx_ctx_create: no HIP device with this index";
    return CPX_ERR_HIP;
  }
  cpx_ctx* c = new (std::nothrow) cpx_ctx();
  if (!c) return CPX_ERR_INTERNAL;
  try {
    c->eng = new Engine(device_id);
  } catch (const std::exception& e) {
    g_create_error = std::string("cpx_ctx_create: ") + e.what();
    delete c;
    return CPX_ERR_HIP;
  } catch (...) {
    delete c;
    return CPX_ERR_HIP;
  }
  *out = c;
  return CPX_OK;
}
void cpx_ctx_destroy(cpx_ctx* ctx) {
  if (!ctx) return;
  delete ctx->eng;
  delete ctx;
}
int cpx_ctx_set_option(cpx_ctx* ctx, const char* key, long long value) {
  if (!ctx || !ctx->eng || !key) return CPX_ERR_ARG;
  if (!ctx->eng->set_option(key, (long)value)) {
    ctx->err = std::string("cpx_ctx_set_option: unknown key or value out of range: ") + key;
    return CPX_ERR_ARG;
  }
  return CPX_OK;
}
int cpx_ctx_get_option(const cpx_ctx* ctx, const char* key, long long* value) {
  if (!ctx || !ctx->eng || !key || !value) return CPX_ERR_ARG;
  long v = 0;
  if (!ctx->eng->get_option(key, &v)) return CPX_ERR_ARG;
  *value = v;
  return CPX_OK;
}
const char* cpx_last_error(const cpx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }   // NULL: why this thread's last cpx_ctx_create failed

int cpx_ctx_set_crs(cpx_ctx* ctx, size_t ell, const uint8_t* points, size_t n_points) {
  if (!points) return CPX_ERR_ARG;
  if (n_points < ell + 7) {   // crs.rs:40-42 "not enough points"
    if (ctx) ctx->err = "CurdleproofsCrs::from_points: not enough points (need ell + 7)";
    return CPX_ERR_ARG;
  }
  return guarded(ctx, [&] {
    ctx->eng->set_crs(ell, points);
    return CPX_OK;
  });
}
int cpx_crs_sums(const cpx_ctx* ctx, uint8_t g_sum[96], uint8_t h_sum[96]) {
  if (!ctx || !ctx->eng || !ctx->eng->ell()) return CPX_ERR_STATE;
  ctx->eng->crs_sums(g_sum, h_sum);
  return CPX_OK;
}
size_t cpx_batch_size(const cpx_ctx* ctx) { return (ctx && ctx->eng) ? ctx->eng->batch() : 0; }
size_t cpx_proof_size(const cpx_ctx* ctx) { return (ctx && ctx->eng && ctx->eng->ell()) ? ctx->eng->proof_size() : 0; }

int cpx_g1_msm(cpx_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[144]) {
  if (!out || (n && (!bases || !scalars))) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    ctx->eng->msm(bases, scalars, n, out);
    return CPX_OK;
  });
}
int cpx_g1_msm_jac(cpx_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[144]) {
  if (!out || (n && (!bases || !scalars))) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    ctx->eng->msm_jac(bases, scalars, n, out);
    return CPX_OK;
  });
}
int cpx_g1_fold(cpx_ctx* ctx, uint8_t* PL, const uint8_t* PR, const uint8_t gamma[32], size_t half) {
  if (half && (!PL || !PR || !gamma)) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    ctx->eng->fold(PL, PR, gamma, half);
    return CPX_OK;
  });
}
int cpx_g1_scale(cpx_ctx* ctx, const uint8_t* P, const uint8_t* scalars, size_t scalar_stride, size_t n, uint8_t* out) {
  if (n && (!P || !scalars || !out)) return CPX_ERR_ARG;
  if (scalar_stride != 0 && scalar_stride != 32) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    ctx->eng->scale(P, scalars, scalar_stride, n, out);
    return CPX_OK;
  });
}
int cpx_g1_normalize(cpx_ctx* ctx, const uint8_t* jac, size_t n, uint8_t* out_affine, uint8_t* out_compressed) {
  if (n && !jac) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    ctx->eng->normalize(jac, n, out_affine, out_compressed);
    return CPX_OK;
  });
}
int cpx_g1_decompress(cpx_ctx* ctx, const uint8_t* compressed, size_t n, uint8_t* out_affine, int check_subgroup) {
  if (n && (!compressed || !out_affine)) return CPX_ERR_ARG;
  return guarded(ctx, [&] { return ctx->eng->decompress(compressed, n, out_affine, check_subgroup); });
}

int cpx_g1_decompress_status(cpx_ctx* ctx, const uint8_t* compressed, size_t n, uint8_t* out_affine, int check_subgroup, uint8_t* status) {
  if (n && (!compressed || !out_affine || !status)) return CPX_ERR_ARG;
  return guarded(ctx, [&] { return ctx->eng->decompress(compressed, n, out_affine, check_subgroup, status); });
}

int cpx_accum_new(cpx_ctx* ctx, cpx_accum** out) {
  if (!ctx || !ctx->eng || !out) return CPX_ERR_ARG;
  cpx_accum* a = new (std::nothrow) cpx_accum();
  if (!a) return CPX_ERR_INTERNAL;
  a->ctx = ctx;
  *out = a;
  return CPX_OK;
}
void cpx_accum_free(cpx_accum* acc) { delete acc; }
int cpx_accum_check(cpx_accum* acc, const uint8_t C[144], const uint8_t* vec_x, const uint8_t* vec_V, size_t n, const uint8_t random_factor[32]) {
  if (!acc || !C || !random_factor || (n && (!vec_x || !vec_V))) return CPX_ERR_ARG;
  return guarded(acc->ctx, [&] {
    host::S a;
    memcpy(a.f.v, random_factor, 32);
    if (!host::is_valid_factor(random_factor)) throw ArgError("random_factor must be a non-zero reduced field element (msm_accumulator.rs:44 draws Fr::rand)");
    acc->lhs_points.insert(acc->lhs_points.end(), C, C + 144);
    acc->lhs_scalars.push_back(a);
    for (size_t i = 0; i < n; i++) {
      cpx_accum::Key k;
      memcpy(k.b, vec_V + 96 * i, 96);
      host::S x;
      memcpy(x.f.v, vec_x + 32 * i, 32);
      auto it = acc->base_scalar.emplace(k, host::S::zero()).first;
      it->second += a * x;
    }
    return CPX_OK;
  });
}
int cpx_accum_verify(cpx_accum* acc) {
  if (!acc) return CPX_ERR_ARG;
  return guarded(acc->ctx, [&] {
    Engine* e = acc->ctx->eng;
    // msm(bases, scalars) - A_c == O, with A_c = sum a_i C_i folded into the same MSM (negated scalars)
    const size_t nl = acc->lhs_scalars.size(), nb = acc->base_scalar.size();
    std::vector<uint8_t> bases((nl + nb) * 96), scalars((nl + nb) * 32);
    if (nl) e->normalize(acc->lhs_points.data(), nl, bases.data(), nullptr);
    for (size_t i = 0; i < nl; i++) {
      host::S m = -acc->lhs_scalars[i];
      memcpy(&scalars[32 * i], m.f.v, 32);
    }
    size_t j = nl;
    for (auto& kv : acc->base_scalar) {
      memcpy(&bases[96 * j], kv.first.b, 96);
      memcpy(&scalars[32 * j], kv.second.f.v, 32);
      j++;
    }
    uint8_t res[144], comp[48];
    e->msm(bases.data(), scalars.data(), nl + nb, res);
    e->normalize(res, 1, nullptr, comp);
    return comp[0] == 0xc0 ? CPX_OK : CPX_ERR_VERIFY;
  });
}

int cpx_batch_load(cpx_ctx* ctx, size_t batch, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T, const uint8_t* vec_U, const uint8_t* M) {
  if (!batch || !vec_R || !vec_S || !vec_T || !vec_U || !M) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    ctx->eng->batch_load(batch, vec_R, vec_S, vec_T, vec_U, M);
    return CPX_OK;
  });
}
int cpx_batch_load_begin(cpx_ctx* ctx, size_t batch, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T, const uint8_t* vec_U, const uint8_t* M) {
  if (!batch || !vec_R || !vec_S || !vec_T || !vec_U || !M) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    ctx->eng->batch_load_begin(batch, vec_R, vec_S, vec_T, vec_U, M);
    return CPX_OK;
  });
}
int cpx_batch_load_end(cpx_ctx* ctx) {
  return guarded(ctx, [&] {
    ctx->eng->batch_load_end();
    return CPX_OK;
  });
}
int cpx_batch_prove(cpx_ctx* ctx, const uint32_t* permutation, const uint8_t* k, const uint8_t* vec_m_blinders, const uint8_t* rand, uint8_t* proofs_out) {
  if (!permutation || !k || !vec_m_blinders || !rand || !proofs_out) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    const size_t B = ctx->eng->batch(), ell = ctx->eng->ell();
    for (size_t i = 0; i < B * ell; i++)
      if (permutation[i] >= ell) return (int)CPX_ERR_ARG;
    ctx->eng->batch_prove(permutation, k, vec_m_blinders, rand, proofs_out);
    return (int)CPX_OK;
  });
}
int cpx_batch_verify(cpx_ctx* ctx, const uint8_t* proofs, const uint8_t* rand, int* verdict) {
  if (!proofs || !rand || !verdict) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    ctx->eng->batch_verify(proofs, rand, verdict);
    return CPX_OK;
  });
}

int cpx_batch_verify_fused(cpx_ctx* ctx, const uint8_t* proofs, const uint8_t* rand, uint8_t* partial_jac, int* n_invalid) {
  if (!proofs || !rand || !partial_jac || !n_invalid) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    ctx->eng->batch_verify_fused(proofs, rand, partial_jac, n_invalid);
    return CPX_OK;
  });
}
int cpx_g1_sum_jac(cpx_ctx* ctx, const uint8_t* points_jac, size_t n, uint8_t* out_jac, int* is_identity) {
  if ((n && !points_jac) || !out_jac || !is_identity) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    *is_identity = ctx->eng->sum_jac(points_jac, n, out_jac) ? 1 : 0;
    return CPX_OK;
  });
}

// ---- Whisk byte-level API (whisk.rs) ----
int cpx_whisk_generate_shuffle_proof(cpx_ctx* ctx, const uint8_t* pre_trackers, const uint32_t* permutation, const uint8_t k[32], const uint8_t* vec_m_blinders,
                                     const uint8_t* rand, uint8_t* post_trackers_out, uint8_t* proof_out) {
  if (!pre_trackers || !permutation || !k || !vec_m_blinders || !rand || !post_trackers_out || !proof_out) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    const size_t ell = ctx->eng->ell();
    std::vector<uint8_t> seen(ell, 0);
    for (size_t i = 0; i < ell; i++) {   // must be a permutation of 0..ell (the reference shuffles (0..ELL))
      if (permutation[i] >= ell || seen[permutation[i]]) return (int)CPX_ERR_ARG;
      seen[permutation[i]] = 1;
    }
    return ctx->eng->whisk_generate_shuffle_proof(pre_trackers, permutation, k, vec_m_blinders, rand, post_trackers_out, proof_out);
  });
}
int cpx_whisk_is_valid_shuffle_proof(cpx_ctx* ctx, const uint8_t* pre_trackers, const uint8_t* post_trackers, const uint8_t* proof, const uint8_t* rand, int* valid) {
  if (!pre_trackers || !post_trackers || !proof || !rand || !valid) return CPX_ERR_ARG;
  return guarded(ctx, [&] { return ctx->eng->whisk_is_valid_shuffle_proof(pre_trackers, post_trackers, proof, rand, valid); });
}
int cpx_whisk_generate_tracker_proof(cpx_ctx* ctx, const uint8_t tracker[96], const uint8_t k[32], const uint8_t blinder[32], uint8_t proof_out[128]) {
  if (!tracker || !k || !blinder || !proof_out) return CPX_ERR_ARG;
  return guarded(ctx, [&] { return ctx->eng->whisk_generate_tracker_proof(tracker, k, blinder, proof_out); });
}
int cpx_whisk_is_valid_tracker_proof(cpx_ctx* ctx, const uint8_t tracker[96], const uint8_t k_commitment[48], const uint8_t proof[128], int* valid) {
  if (!tracker || !k_commitment || !proof || !valid) return CPX_ERR_ARG;
  return guarded(ctx, [&] { return ctx->eng->whisk_is_valid_tracker_proof(tracker, k_commitment, proof, valid); });
}

int cpx_set_profiling(cpx_ctx* ctx, int on) {
  if (!ctx || !ctx->eng) return CPX_ERR_ARG;
  ctx->eng->set_profiling(on != 0);
  return CPX_OK;
}
int cpx_reset_stats(cpx_ctx* ctx) {
  if (!ctx || !ctx->eng) return CPX_ERR_ARG;
  ctx->eng->reset_stats();
  return CPX_OK;
}
int cpx_get_stat(const cpx_ctx* ctx, const char* name, uint64_t* launches, double* total_ms, double* algorithmic_bytes, double* units) {
  if (!ctx || !ctx->eng || !name) return CPX_ERR_ARG;
  auto& m = ctx->eng->stats();
  auto it = m.find(name);
  KernelStat st;
  if (it != m.end()) st = it->second;
  if (launches) *launches = st.launches;
  if (total_ms) *total_ms = st.ms;
  if (algorithmic_bytes) *algorithmic_bytes = st.alg_bytes;
  if (units) *units = st.units;
  return CPX_OK;
}
int cpx_set_host_threads(cpx_ctx* ctx, int threads) {
  if (!ctx || !ctx->eng) return CPX_ERR_ARG;
  ctx->eng->set_host_threads(threads);
  return CPX_OK;
}
int cpx_bench_fpmul(cpx_ctx* ctx, int blocks, int iters, int reps, double* products_per_second) {
  if (!products_per_second || blocks <= 0 || iters <= 0 || reps <= 0) return CPX_ERR_ARG;
  return guarded(ctx, [&] {
    *products_per_second = ctx->eng->bench_fpmul(blocks, iters, reps);
    return CPX_OK;
  });
}

}  // extern "C"
