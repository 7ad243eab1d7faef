// Host engine of the MI355X Curdleproofs core — product code.
//
// One `Engine` = one HIP device + one stream + the device-resident CRS and batch buffers.  A batch of
// B independent shuffle instances (same ell) advances in lock-step: every protocol phase is ONE set of
// kernel launches covering all B proofs (the batch is a grid dimension), bracketed by the host-side
// Fiat-Shamir step.  Mirrors /root/reference/src/curdleproofs.rs:59-298 and the sub-arguments it calls.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/cpx.h"   // public error codes
#include "host_math.hpp"
#include "host_threads.hpp"
#include "kernels.h"
#include "protocol.h"

namespace cpx {

struct HipError : std::runtime_error {
  hipError_t code;
  HipError(hipError_t c, const char* what) : std::runtime_error(what), code(c) {}
};
// a caller-supplied value is out of range (-> CPX_ERR_ARG)
struct ArgError : std::runtime_error {
  explicit ArgError(const char* what) : std::runtime_error(what) {}
};
#define CPX_HIP(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) throw ::cpx::HipError(_e, (std::string(#expr) + ": " + hipGetErrorString(_e)).c_str()); \
  } while (0)

// Page-locked host memory comes from the (uninstrumented) HIP runtime, which hands a freed range out again: ThreadSanitizer cannot see the
// runtime's lock between one owner's hipHostFree and the next owner's hipHostMalloc of the same addresses — one context's staging buffer
// released, another context's status buffer allocated there — and reports the two owners' accesses as a race (tests/test_sanitizers.py, the
// engine under TSan on the GPU box).  The free / allocate pair is therefore announced to the tool as a release / acquire on one
// process-wide object; a no-op in every other build.
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define CPX_TSAN 1
#endif
#endif
#if defined(CPX_TSAN)
extern "C" void AnnotateHappensBefore(const char* file, int line, const volatile void* addr);
extern "C" void AnnotateHappensAfter(const char* file, int line, const volatile void* addr);
inline char g_pinned_memory_sync = 0;
#define CPX_HOST_MEMORY_RELEASE() AnnotateHappensBefore(__FILE__, __LINE__, &::cpx::g_pinned_memory_sync)
#define CPX_HOST_MEMORY_ACQUIRE() AnnotateHappensAfter(__FILE__, __LINE__, &::cpx::g_pinned_memory_sync)
#else
#define CPX_HOST_MEMORY_RELEASE() ((void)0)
#define CPX_HOST_MEMORY_ACQUIRE() ((void)0)
#endif

template <class T> struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  void ensure(size_t n) {
    if (n <= cap) return;
    if (p) CPX_HIP(hipFree(p));
    p = nullptr;
    CPX_HIP(hipMalloc(&p, n * sizeof(T)));
    cap = n;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};
template <class T> struct PinBuf {
  T* p = nullptr;
  size_t cap = 0;
  void ensure(size_t n) {
    if (n <= cap) return;
    CPX_HOST_MEMORY_RELEASE();
    if (p) CPX_HIP(hipHostFree(p));
    p = nullptr;
    CPX_HIP(hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault));
    CPX_HOST_MEMORY_ACQUIRE();
    cap = n;
  }
  ~PinBuf() {
    CPX_HOST_MEMORY_RELEASE();
    if (p) (void)hipHostFree(p);
  }
};

struct KernelStat {
  uint64_t launches = 0;
  double ms = 0;
  double alg_bytes = 0;   // algorithmic bytes moved (SURVEY §8d accounting)
  double units = 0;       // MSM points / scalar-mul elements / points processed
};


// kernel names as rocprofv3 reports the template instantiations (statistics keys)
inline const char* fix_kernel_name(int bits, int wpw) {
  if (bits == 19) return "k_msm_fix<19, 7>";
  if (bits == 16) return wpw == 16 ? "k_msm_fix<16, 16>" : wpw == 8 ? "k_msm_fix<16, 8>" : wpw == 4 ? "k_msm_fix<16, 4>" : "k_msm_fix<16, 2>";
  return wpw == 16 ? "k_msm_fix<8, 16>" : "k_msm_fix<8, 8>";
}
inline const char* tblw_kernel_name(int wpw) {
  switch (wpw) {
    case 32: return "k_msm_tblw<32, false>";
    case 16: return "k_msm_tblw<16, false>";
    case 8: return "k_msm_tblw<8, false>";
    case 4: return "k_msm_tblw<4, false>";
    default: return "k_msm_tblw<2, false>";
  }
}


void trace_scalar(const char* name, const Fr& x);   // CPX_TRACE=1 debugging aid (engine_device.cpp)

class Engine {
 public:
  explicit Engine(int device);
  ~Engine();

  // crs.rs:37-58 CurdleproofsCrs::from_points: ell + 7 affine points (G[ell] | H[4] | H | G_t | G_u)
  void set_crs(size_t ell, const uint8_t* points);
  size_t ell() const { return ell_; }
  size_t n() const { return n_; }
  size_t log2n() const { return L_; }
  size_t proof_size() const { return 48 * (18 + 10 * L_) + 32 * 7; }
  void crs_sums(uint8_t* g_sum, uint8_t* h_sum) const;

  // ---- tier 0: the reference's util::msm & friends on caller buffers ----
  void msm(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t* out_jac);                  // util.rs:19-22
  void msm_jac(const uint8_t* bases_jac, const uint8_t* scalars, size_t n, uint8_t* out_jac);          // util.rs:25-29
  void fold(uint8_t* PL, const uint8_t* PR, const uint8_t* gamma, size_t half);                        // IPA / SameMSM folds
  void scale(const uint8_t* P, const uint8_t* scalars, size_t scalar_stride, size_t n, uint8_t* out);  // G' rescale / k*R
  void normalize(const uint8_t* jac, size_t n, uint8_t* out_aff, uint8_t* out_comp);                   // normalize_batch (+compress)
  int decompress(const uint8_t* comp, size_t n, uint8_t* out_aff, int check_subgroup, uint8_t* status_out = nullptr);   // whisk.rs:318-320

  // ---- tier 2: batches of whole proofs, instance data resident in HBM ----
  void batch_load(size_t batch, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T, const uint8_t* vec_U, const uint8_t* M);
  // the next batch's instance uploaded beside the running batch (include/cpx.h cpx_batch_load_begin / _end)
  void batch_load_begin(size_t batch, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T, const uint8_t* vec_U, const uint8_t* M);
  void batch_load_end();
  // witnesses + the 3n+9 Fr draws per proof (SURVEY §8b RNG contract); writes batch * proof_size() bytes
  void batch_prove(const uint32_t* permutation, const uint8_t* k, const uint8_t* m_blinders, const uint8_t* rand, uint8_t* proofs_out);
  // proofs: batch * proof_size() bytes; rand: batch * 8 Fr; verdict[i] = CPX_OK / CPX_ERR_VERIFY / CPX_ERR_DESERIALIZE
  bool sum_jac(const uint8_t* points_jac, size_t n, uint8_t* out_jac);   // true iff the sum is the identity
  void batch_verify(const uint8_t* proofs, const uint8_t* rand, int* verdict);
  // BASELINE config 5: all proofs of the batch in one accumulated MSM; 12 random factors per proof; output = this
  // engine's partial sum (Jacobian, standard form) and the number of structurally invalid proofs
  void batch_verify_fused(const uint8_t* proofs, const uint8_t* rand, uint8_t* partial_jac, int* n_invalid);
  size_t batch() const { return B_; }

  // ---- Whisk byte-level API (whisk.rs; whisk.cpp) ----
  int whisk_generate_shuffle_proof(const uint8_t* pre_trackers, const uint32_t* permutation, const uint8_t* k, const uint8_t* vec_m_blinders, const uint8_t* rand,
                                   uint8_t* post_trackers_out, uint8_t* proof_out);
  int whisk_is_valid_shuffle_proof(const uint8_t* pre_trackers, const uint8_t* post_trackers, const uint8_t* proof, const uint8_t* rand, int* valid);
  int whisk_generate_tracker_proof(const uint8_t tracker[96], const uint8_t k[32], const uint8_t blinder[32], uint8_t proof_out[128]);
  int whisk_is_valid_tracker_proof(const uint8_t tracker[96], const uint8_t k_commitment[48], const uint8_t proof[128], int* valid);

  // ---- measurement ----
  void set_profiling(bool on) { profiling_ = on; }
  void reset_stats() { stats_.clear(); }
  const std::map<std::string, KernelStat>& stats() const { return stats_; }
  double bench_fpmul(int blocks, int iters, int reps);   // returns Fp products per second
  void set_host_threads(int t) {
    host_threads_ = t;
    pool_.reset();
  }
  // per-context tunables (kernels.h `Options`; cpx_ctx_set_option).  fix_bits takes effect at the next set_crs; a change invalidates
  // the cached device plans (their task layouts depend on the kernel selection).
  bool set_option(const char* key, long value);
  // read-only beside the tunables: "fix_bits_effective" = the radix of the fixed-base table in use (set_crs falls back 19 -> 16 -> 8 when
  // free HBM is short; 0 before the first set_crs)
  bool get_option(const char* key, long* value) const {
    if (key && !strcmp(key, "fix_bits_effective")) {
      if (value) *value = crs_tab_ ? fix_bits_ : 0;
      return true;
    }
    return cpx::get_option(opt_, key, value);
  }
  hipStream_t stream() const { return stream_; }
  void sync() { CPX_HIP(hipStreamSynchronize(stream_)); }

 private:
  struct MsmReq {
    const Aff* bases;
    const uint32_t* idx;
    const host::S* scalars;   // host pointer, n entries
    uint32_t n;
    uint32_t dst;             // index into d_pp_ (Aff units) receiving the affine result
  };
  // table-backed MSM request: up to two base segments, each with its own host scalar vector
  struct TblReq {
    TblSeg seg0;
    const host::S* s0;
    TblSeg seg1;
    const host::S* s1;
    uint32_t dst = ~0u;       // optional: index into d_pp_ receiving the affine result
    const Fr* dev = nullptr;  // optional: the seg0.n + seg1.n scalars already sit in device memory (s0 / s1 unused)
    uint32_t add[3] = {~0u, ~0u, ~0u};   // optional: d_pp_ indices of affine points of earlier phases added with coefficient 1
  };
  // second stream + private staging for work that is off the critical path of the phase sequence
  struct SideBufs {
    hipStream_t stream = nullptr;
    hipStream_t hi_stream = nullptr;   // high-priority stream: the transcript prefix of a small batch (its waves are placed before those of the table build)
    hipStream_t lat_stream = nullptr, lat_main = nullptr;   // the host-driven prover's side / main stream: CU-masked (create_masked_stream)
    hipEvent_t lat_ev = nullptr;
    hipEvent_t ev = nullptr, ev2 = nullptr;
    DevBuf<MsmTask> tasks;
    DevBuf<SmulTask> stasks;
    DevBuf<Fr> scal;
    DevBuf<TJac> part, mid;
    DevBuf<TblTask> ttasks;
    DevBuf<uint32_t> raw, rawslot, digits;
    DevBuf<TAff> conv;
    DevBuf<Jac> res;
    DevBuf<uint32_t> dst;
    DevBuf<uint8_t> comp;
    PinBuf<uint8_t> stage, hcomp;
  };
  SideBufs side_;
  // third stream of the host-driven prover: the per-proof tables of T and U and the two commitments that need them first (B_t, B_u of
  // SameMSM step 1) are not on the critical path of a lone proof until the SameMSM transcript step — they run here, beside phases 1..3
  // and the IPA rounds
  struct TabBufs {
    hipStream_t stream = nullptr;
    hipStream_t dstream = nullptr;   // the device-resident prover's table stream (plain): table build + B_t, B_u beside phase 1 (engine_device.cpp)
    hipEvent_t ev_start = nullptr, ev_m = nullptr, ev_done = nullptr;   // M finalised (main) / M's table row built / B_t, B_u compressed
    DevBuf<uint8_t> blob;      // vec_r scalars | TblTask | first | count | dst
    PinBuf<uint8_t> stage;
    DevBuf<uint32_t> raw, rawslot;
    DevBuf<TJac> mid, part;
    DevBuf<uint8_t> comp;
    PinBuf<uint8_t> hcomp;
  };
  TabBufs tab_;
  struct Timed {
    hipEvent_t a, b;
    std::string name;
    double bytes, units;
  };

  void set_crs_impl(size_t ell, const uint8_t* points);
  hipStream_t create_masked_stream(bool upper);
  hipStream_t prefix_stream();
  bool table_stream_on() const;
  const Aff& generator();
  void compress_affine(const Aff* pts, size_t n, uint8_t* out);
  bool unzip_trackers(const uint8_t* trackers, size_t n, std::vector<Aff>& vec_r, std::vector<Aff>& vec_s);
  Aff gen_;
  bool have_gen_ = false;
  std::vector<Aff> crs_host_;   // the ell + 7 CRS points (host copy: M of the Whisk shuffle is an MSM over vec_G | vec_H)
  // ---- device-resident batch prover / verifier (engine_device.cpp) ----
  struct TblPlan {   // task descriptors of one table-backed MSM phase whose scalars live in device memory, built once
    size_t nt = 0, ntt = 0, nft = 0, nparts = 0, fix_sets = 0, tbl_sets = 0;
    int fix_wpw = 16, tbl_wpw = 32;
    double pts_fix = 0, pts_tbl = 0;
    bool any_add = false;
    bool table_stream = false;   // runs beside the main stream's plans (phase 1t): its own write-only dummy slot
    bool combined = false;       // fused SameMSM rounds: wave w of a proof runs fixed-base block w and table block w (RoundDev::combine)
    bool keep_order = false;     // tasks stay in request (= proof-major) order: the fused round kernels address them by proof (round.hip)
    int force_fix_wpw = 0, force_tbl_wpw = 0;   // != 0: the windows per wave instead of the launch-size heuristics
    DevBuf<TblTask> ttasks;
    DevBuf<FixTask> ftasks;
    DevBuf<uint32_t> meta;   // first partial | partial count | affine destination | compressed-bytes slot | addends[3], per request
  };
  struct DevProver {
    std::vector<const void*> signature;   // buffer addresses + shape the plans were built for
    ProveDev dev;
    DevBuf<uint32_t> perm, mdst;
    DevBuf<Fr> k, mbl, rnd, vec, sc, rvec2;
    DevBuf<uint8_t> slotcomp, proofs;
    TblPlan p1, p1b, p1t, p2, p3;   // p1t: B_t, B_u — the two commitments of phase 1 over per-proof tables, on the table stream
    std::vector<std::unique_ptr<TblPlan>> ipa, smsm;
    // fused log rounds (round.hip; options fused_rounds_max, fused_fix_wpw, fused_tbl_wpw): the plans above in proof-major order, one launch per round
    bool fused = false, fused_smsm = false;   // the IPA rounds / the SameMSM rounds (and the small-batch form of phase 2)
    DevBuf<TAcc> rpart;          // XYZZ partial sums of a round's MSM waves
    DevBuf<uint32_t> rcount;     // [B] arrival counters
    // the last log rounds on materialised folded bases (late.hip; options late_rounds, late_min_batch)
    struct Late {
      bool on = false;
      int m = 16, nr = 4;               // materialised bases per family, late rounds = log2(m)
      size_t j0 = 0;                    // first late round
      DevBuf<TJac> jac;                 // [LATE_FAMILIES][B][m] materialised points
      DevBuf<TAff> tab;                 // [LATE_FAMILIES][B][m][late_tab_entries()] their small multiples
      DevBuf<TJac> part, extra;         // window-group partial sums of a round's outputs [6 B][slices]; the IPA rounds' CRS terms [4 B]
      const uint32_t* h_col = nullptr;
      DevBuf<uint32_t> meta;            // iota | ones | compressed-bytes slots of every late round's outputs
      std::vector<LateRound> ipa, smsm;
      std::vector<size_t> ipa_comp, smsm_comp;   // offsets of a round's slot list inside meta
      const uint32_t* gb_cols = nullptr;
    } late;
    const uint32_t* side_cols = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr, ev_t1 = nullptr, ev_t2 = nullptr, ev_a2 = nullptr;   // t1: tables built, t2: B_t, B_u in the registry, a2: the prefix stream is about to launch
  };
  DevProver dprove_;
  // Scratch of the tier-0 calls (msm, fold, scale, normalize, decompress, sum_jac): kept between calls — a call used to allocate and free up to ten
  // device buffers of its own (hipMalloc + hipFree: 1.9 ms per 256-point cpx_g1_msm, most of it the allocator); a buffer that has grown beyond
  // kTier0Keep bytes for one large call is given back at the end of that call.
  struct Tier0 {
    DevBuf<Aff> a0, a1;
    DevBuf<Fr> fr;
    DevBuf<MsmTask> mtask;
    DevBuf<SmulTask> stask;
    DevBuf<TJac> w, pt, part;
    DevBuf<Jac> res, jin;
    DevBuf<TAff> conv;
    DevBuf<TblTask> ttask;
    DevBuf<uint32_t> dig;
    DevBuf<uint8_t> bytes, status;
    DevBuf<int> flag;
  } t0_;
  static constexpr size_t kTier0Keep = (size_t)64 << 20;
  template <class T> static void tier0_trim(DevBuf<T>& b) {
    if (b.cap * sizeof(T) > kTier0Keep) b.release();
  }
  void enqueue_prove_device();
  void exec_late_round(const LateRound& r, size_t comp_off, const char* what);
  // proofs per launch of the device prover's table build (option table_chunks).  Default: ALL rows in one launch while its scratch (15 doubled
  // copies of every base: 13.5 GB at 8192 proofs of ell = 252) stays below 16 GiB, equal chunks of that size above.  Round 5 shipped two chunks of
  // 4096 (6.7 GB of scratch) on a cross-box comparison; the in-process A/B of round 6 (bench.py --ab table_chunks=0,1, profiles/r06_ab_table_chunks.json)
  // has all rows 0.53 % faster in six of six rounds (every chunk launch ends in a tail of long waves) — and HBM footprint is not the metric.
  size_t table_chunk_rows(size_t batch) const {
    const size_t bytes = batch * np() * (size_t)(copies_ / 2 - 1) * sizeof(TblTmp);
    const size_t nch = opt_.table_chunks > 0 ? std::min<size_t>((size_t)opt_.table_chunks, std::max<size_t>(1, batch))
                                             : std::max<size_t>(1, (bytes + ((size_t)16 << 30) - 1) >> 34);
    return (batch + nch - 1) / nch;
  }
  struct DevVerifier {
    std::vector<const void*> signature;
    VerifyDev dev;
    DevBuf<uint8_t> proofs, slotcomp, status;
    DevBuf<Fr> rnd, vsc, scal, scal_crs;
    DevBuf<uint32_t> flags, src_off, dst, mdst, gidx;
    DevBuf<MsmTask> mtasks, gtasks;
    DevBuf<FixTask> ftasks;
    TblPlan pd;                                  // D and A'
    int fix_wpw = 16, fix_parts = 1, fix_wpw1 = 2, fix_parts1 = 8;
    size_t G = 1, NT = 1;                        // fused batch: proofs per group, groups
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
  };
  DevVerifier dverify_;
  void prepare_device_verifier(size_t rand_stride);
  void verify_core_device(const uint8_t* proofs, const uint8_t* rand, size_t rand_stride, int* verdict, uint8_t* fused_partial, int* fused_invalid);
  void build_plan(TblPlan& pl, const std::vector<TblReq>& reqs, const std::vector<uint32_t>& comp_index);
  void exec_plan(const TblPlan& pl, uint8_t* d_comp_registry, bool on_table_stream = false);
  void prepare_device_prover();
  void batch_prove_device(const uint32_t* permutation, const uint8_t* k, const uint8_t* m_blinders, const uint8_t* rand, uint8_t* proofs_out);

  void run_msm_phase(const std::vector<MsmReq>& reqs, std::vector<uint8_t>* comp_out);
  void run_tbl_phase(const std::vector<TblReq>& reqs, std::vector<uint8_t>* comp_out);
  void ensure_raw_sets(size_t nsets);
  void reduce_sets(size_t nplain, size_t nweighted, TJac* part = nullptr);
  void verify_core(const uint8_t* proofs, const uint8_t* rand, size_t rand_stride, int* verdict, uint8_t* fused_partial, int* fused_invalid);
  void batch_prove_tables(const uint32_t* permutation, const uint8_t* k, const uint8_t* m_blinders, const uint8_t* rand, uint8_t* proofs_out);
  // option strict_infinity = 0: `bytes` with every non-canonical infinity encoding (at `offsets` of each of the nrec records) rewritten to
  // the canonical one, in canon_bytes_ (valid until the next call); `bytes` itself when there is none or the option is set
  const uint8_t* canonical_infinities(const uint8_t* bytes, size_t nbytes, size_t nrec, size_t rec_stride, const std::vector<size_t>& offsets);
  std::vector<uint8_t> canon_bytes_;
  void run_smul(const std::vector<SmulTask>& tasks, int cnt, const host::S* scalars, size_t nscalars, double alg_bytes);
  const uint32_t* idx_list(const std::vector<uint32_t>& v);
  void tick(const char* name, double bytes, double units, bool span = false);
  bool span_ = false;
  void tock();
  void flush_timers();
  template <class F> void parallel_for(size_t n, F&& f);
  std::unique_ptr<SpinTeam> team_;
  static std::atomic<int>& live_engines();   // engine contexts alive in this process (bounds the spin team)
  struct TeamScope {   // engages the spin team for a small batch's call
    SpinTeam* t = nullptr;
    TeamScope(Engine* e, size_t batch);
    ~TeamScope() {
      if (t) t->release();
    }
  };
  struct HostSpan {
    HostSpan(Engine* e, const char* name);
    ~HostSpan();
    Engine* e_;
    const char* name_;
    std::chrono::steady_clock::time_point t0_;
  };
  void wait_stream();
  void wait_stream_blocking();                // sleeps instead of spinning: the one wait of a device-resident call lasts tens of milliseconds
  hipEvent_t ev_block_ = nullptr;
  void wait_side();                          // blocks on the side stream's event (side_.ev)
  void transcript_prefix_async(size_t B);
  bool device_prefix(size_t B) const;         // run the whole protocol on the GPU (batches >= CPX_DEVICE_MIN_BATCH) or drive it from the host

  // per-proof table row: copy-major [copies][NP]
  size_t np() const { return 1 + 2 * n_; }   // M | T_b | U_b   (R and S are used once: no table)
  TAff* ptab(size_t p) const { return d_ptab_.p + p * (size_t)copies_ * np(); }
  TblSeg pseg(size_t p, size_t off, uint32_t cnt, const uint32_t* idx = nullptr) const { return TblSeg{ptab(p) + off, idx, (uint32_t)np(), cnt}; }
  TblSeg cseg(size_t off, uint32_t cnt, const uint32_t* idx = nullptr) const { return TblSeg{ctab() + off, idx, (uint32_t)nc(), cnt}; }
  size_t nc() const { return n_ + 5; }   // CRS table columns: G | Hvec | H | G_t | G_u | G_sum | H_sum
  Aff* pp(size_t p) const { return d_pp_.p + p * pp_stride_; }
  Aff* slot(size_t p, int s) const { return pp(p) + 4 * ell_ + s; }
  uint32_t slot_index(size_t p, int s) const { return (uint32_t)(p * pp_stride_ + 4 * ell_ + s); }

  int device_;
  Options opt_ = default_options();
  hipStream_t stream_ = nullptr;
  bool profiling_ = false;
  int host_threads_ = 0;
  std::unique_ptr<WorkerPool> pool_;
  std::map<std::string, KernelStat> stats_;
  std::vector<Timed> pending_;

  // CRS
  size_t ell_ = 0, n_ = 0, L_ = 0;
  DevBuf<Aff> d_crs_;        // [n+1] G | Hvec | H      (IPA basis + H)
  DevBuf<Aff> d_crs_gb_;     // [n]   G | Hvec[0..2) | G_t | G_u   (SameMSM basis)
  Aff crs_single_[5];        // H, G_t, G_u, G_sum, H_sum (host copies)
  uint8_t crs_H_comp_[48];

  // shifted-base tables (all-MSM prover)
  static constexpr int copies_ = 32;   // table copies per base: 2^(8c) P and z^2 2^(8c) P, c < 16: one per radix-256 window of the split scalar
  // CRS tables are immutable once built and large (15 GB at ell = 252): engines on the same device that are
  // given the same CRS share one copy (process-wide registry in engine.cpp).
  struct CrsTables {
    int device = 0, fix_bits = 0;
    std::vector<uint8_t> key;   // the CRS points the tables were built from
    DevBuf<TAff> ctab;         // shifted copies [copies][n+3] : G | Hvec | H | G_t | G_u   (table representation)
    DevBuf<TFix> fixtab;       // multiples [256/c][2^(c-1)][n+5], one 128-byte line per entry
  };
  std::shared_ptr<CrsTables> crs_tab_;
  const TAff* ctab() const { return crs_tab_ ? crs_tab_->ctab.p : nullptr; }
  const TFix* fixtab() const { return crs_tab_ ? crs_tab_->fixtab.p : nullptr; }
  DevBuf<TAff> d_ptab_;      // per-proof tables [B][copies][NP]
  DevBuf<uint32_t> d_raw_, d_rawslot_;   // raw lane accumulators of k_msm_tblw / k_msm_fix and their partial-sum slots
  DevBuf<TJac> d_rawmid_;                // group sums between the two reduction launches
  DevBuf<uint32_t> d_digits_;            // recoded scalars of the endomorphism bucket MSM (9 words per point)
  DevBuf<Aff> d_psrc_;       // their standard-form sources [B][NP] : M | T_b | U_b
  DevBuf<TblTmp> d_tbltmp_;
  DevBuf<TblTask> d_ttasks_;
  int fix_bits_cfg_ = 16;    // configured radix of the fixed-base CRS table of multiples: 2^16 (15 GB at ell = 252) or 2^8 (0.1 GB)
  int fix_bits_ = 16;        // radix of the table in use (set_crs falls back to 2^8 when the 2^16 table does not fit in free HBM)
  DevBuf<FixTask> d_ftasks_;
  DevBuf<uint8_t> d_blob_;   // scalars + task descriptors + ranges of a host-driven table phase, one upload (run_tbl_phase)

  // batch
  size_t B_ = 0;
  size_t consts_rows_ = 0;   // rows of d_pp_ / d_psrc_ whose per-CRS constants are in place (load_rows)
  void load_rows(size_t batch, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T, const uint8_t* vec_U, const uint8_t* M, bool from_device);
  struct Stage {   // device staging area of batch_load_begin
    DevBuf<Aff> R, S, T, U;
    DevBuf<Jac> M;
    hipStream_t stream = nullptr;
    hipEvent_t uploaded = nullptr, consumed = nullptr;
    bool have_consumed = false;
    size_t batch = 0, ell = 0;
  } stage_;
  size_t pp_stride_ = 0;     // 4*ell + NSLOT
  DevBuf<Aff> d_pp_;         // per proof: R|S|T|U | slots
  DevBuf<Jac> d_Mjac_;

  // phase scratch
  DevBuf<MsmTask> d_tasks_;
  DevBuf<SmulTask> d_stasks_;
  DevBuf<Fr> d_scal_;
  DevBuf<TJac> d_wsum_, d_part_;   // window sums / Horner partials of the bucket MSM (table representation)
  DevBuf<Jac> d_res_;
  DevBuf<uint32_t> d_big_idx_;   // gather list of the fused verifier's per-proof points
  DevBuf<TAff> d_conv_;            // table-form copies of the bases of a bucket-MSM phase
  DevBuf<TJac> d_tpart_;           // partial sums of the table kernels
  DevBuf<uint32_t> d_dst_;
  DevBuf<uint8_t> d_comp_;
  PinBuf<uint8_t> h_stage_;
  PinBuf<uint8_t> h_comp_;
  PinBuf<uint8_t> h_inst_comp_, h_mcomp_, h_pts_, h_status_;   // compressed instance vectors / M, proof points in, decompression status
  PinBuf<uint32_t> h_u32_;
  DevBuf<Fr> d_rvec_, d_rout_, d_rgam_, d_rbeta_;   // device-resident round vectors, per-round scalars, challenges
  PinBuf<Fr> h_rvec_, h_rgam_, h_rfin_;
  DevBuf<uint8_t> d_mcomp_;                 // compressed M of every proof
  DevBuf<uint64_t> d_tstate_;               // transcript states after the prefix [B][27]
  DevBuf<Fr> d_veca_;                       // vec_a [B][ell]
  DevBuf<uint8_t> d_vin_;
  DevBuf<uint32_t> d_vdst_;
  std::map<std::vector<uint32_t>, uint32_t*> idx_cache_;
  std::vector<uint32_t*> idx_allocs_;
  DevBuf<uint8_t> d_bytes_;
  DevBuf<uint8_t> d_status_;
};

}  // namespace cpx
