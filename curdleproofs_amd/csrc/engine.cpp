// Host engine of the MI355X Curdleproofs core — see engine.hpp.  Product code: no CPU fallback for
// the group arithmetic exists here; every point operation below is a kernel launch (kernels.hip).
#include "engine.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace cpx {

using host::S;
using host::SVec;
using host::Transcript;

static const size_t N_BLINDERS = 4;   // /root/reference/src/lib.rs:35

static inline S S_from_wire(const uint8_t* b) {
  S s;
  memcpy(s.f.v, b, 32);
  return s;
}

std::atomic<int>& Engine::live_engines() {
  static std::atomic<int> n{0};
  return n;
}
Engine::Engine(int device) : device_(device) {
  live_engines()++;
  try {
    CPX_HIP(hipSetDevice(device_));
    CPX_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    CPX_HIP(hipStreamCreateWithFlags(&side_.stream, hipStreamNonBlocking));
    CPX_HIP(hipEventCreateWithFlags(&side_.ev, hipEventDisableTiming));
  } catch (...) {   // a half-built engine owns nothing afterwards
    if (side_.ev) (void)hipEventDestroy(side_.ev);
    if (side_.stream) (void)hipStreamDestroy(side_.stream);
    if (stream_) (void)hipStreamDestroy(stream_);
    live_engines()--;
    throw;
  }
  fix_bits_cfg_ = fix_bits_ = (int)opt_.fix_bits;   // radix of the fixed-base CRS table (8: 0.1 GB, 16: 17.5 GB at ell = 252)
}
bool Engine::set_option(const char* key, long value) {
  if (!cpx::set_option(opt_, key, value)) return false;
  fix_bits_cfg_ = (int)opt_.fix_bits;   // used by the next set_crs
  dprove_.signature.clear();            // plans are laid out for a kernel selection: rebuild them
  dverify_.signature.clear();
  return true;
}
Engine::~Engine() {
  live_engines()--;
  for (auto p : idx_allocs_) (void)hipFree(p);
  if (side_.ev) (void)hipEventDestroy(side_.ev);
  if (side_.ev2) (void)hipEventDestroy(side_.ev2);
  if (ev_block_) (void)hipEventDestroy(ev_block_);
  for (hipEvent_t e : {tab_.ev_start, tab_.ev_m, tab_.ev_done})
    if (e) (void)hipEventDestroy(e);
  if (tab_.stream) (void)hipStreamDestroy(tab_.stream);
  if (tab_.dstream) (void)hipStreamDestroy(tab_.dstream);
  for (hipEvent_t e : {dprove_.ev_a, dprove_.ev_b, dprove_.ev_c, dprove_.ev_d, dprove_.ev_t1, dprove_.ev_t2, dprove_.ev_a2, dverify_.ev_a, dverify_.ev_b})
    if (e) (void)hipEventDestroy(e);
  if (stage_.uploaded) (void)hipEventDestroy(stage_.uploaded);
  if (stage_.consumed) (void)hipEventDestroy(stage_.consumed);
  if (stage_.stream) (void)hipStreamDestroy(stage_.stream);
  if (side_.stream) (void)hipStreamDestroy(side_.stream);
  if (side_.hi_stream) (void)hipStreamDestroy(side_.hi_stream);
  if (side_.lat_stream) (void)hipStreamDestroy(side_.lat_stream);
  if (side_.lat_main) (void)hipStreamDestroy(side_.lat_main);
  if (side_.lat_ev) (void)hipEventDestroy(side_.lat_ev);
  if (stream_) (void)hipStreamDestroy(stream_);
}

// ---------------------------------------------------------------- timing
void Engine::tick(const char* name, double bytes, double units, bool span) {
  if (!profiling_) return;
  Timed t;
  t.units = units;
  CPX_HIP(hipEventCreate(&t.a));
  CPX_HIP(hipEventCreate(&t.b));
  t.name = name;
  t.bytes = bytes;
  span_ = span;
  if (span) CPX_HIP(hipEventRecord(t.a, stream_));   // a sequence of launches: bracket it on the stream
  else set_launch_events(t.a, t.b);                  // one dispatch: events bound to kernel begin / end
  pending_.push_back(t);
}
void Engine::tock() {
  if (!profiling_) return;
  if (span_) {
    CPX_HIP(hipEventRecord(pending_.back().b, stream_));
  } else if (launches_since_set() == 0) {   // nothing was launched (empty phase): give the events a defined state
    CPX_HIP(hipEventRecord(pending_.back().a, stream_));
    CPX_HIP(hipEventRecord(pending_.back().b, stream_));
  }
  set_launch_events(nullptr, nullptr);
  span_ = false;
}
void Engine::flush_timers() {
  if (pending_.empty()) return;
  CPX_HIP(hipStreamSynchronize(stream_));
  for (auto& t : pending_) {
    float ms = 0;
    CPX_HIP(hipEventElapsedTime(&ms, t.a, t.b));
    KernelStat& st = stats_[t.name];
    st.launches++;
    st.ms += ms;
    st.alg_bytes += t.bytes;
    st.units += t.units;
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  pending_.clear();
}

Engine::TeamScope::TeamScope(Engine* e, size_t batch) {
  if (!e->opt_.spin_team || batch < 2) return;
  const size_t team_max = (size_t)e->opt_.spin_team_threads;   // size of the team when the context's host-thread count is not set
  // busy-waiting helpers only pay while they have cores of their own: the team is bounded by the cores this process may use
  // (affinity mask and cgroup CPU quota) divided by the engine contexts alive in the process
  const size_t fair = std::max<size_t>(1, effective_host_cores() / std::max<size_t>(1, (size_t)live_engines().load()));
  const size_t want = std::min({batch, e->host_threads_ > 0 ? (size_t)e->host_threads_ : team_max, fair});
  if (want < 2) return;
  if (!e->team_ || e->team_->size() < want) e->team_.reset(new SpinTeam(want - 1));
  t = e->team_.get();
  t->engage();
}

template <class F> void Engine::parallel_for(size_t n, F&& f) {
  if (team_ && team_->engaged() && n >= 2) {   // a small batch's call: spinning helpers (TeamScope)
    const std::function<void(size_t)> fn = [&](size_t i) { f(i); };
    HostSpan w(this, "host_parallel_for");
    team_->run(n, fn);
    return;
  }
  const size_t inline_below = (size_t)opt_.inline_below;
  if (n < inline_below) {   // a handful of items: waking the pool (tens of sleeping threads) costs more than the work
    for (size_t i = 0; i < n; i++) f(i);
    return;
  }
  if (!pool_) {
    size_t T = host_threads_ > 0 ? (size_t)host_threads_ : std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 64);
    pool_.reset(new WorkerPool(T));
  }
  const std::function<void(size_t)> fn = [&](size_t i) { f(i); };
  HostSpan w(this, "host_parallel_for");
  pool_->run(n, fn);
}

// wall-clock spans of host work, reported next to the kernel statistics when profiling is on
Engine::HostSpan::HostSpan(Engine* e, const char* name) : e_(e), name_(name), t0_(std::chrono::steady_clock::now()) {}
Engine::HostSpan::~HostSpan() {
  if (!e_->profiling_) return;
  KernelStat& st = e_->stats_[name_];
  st.launches++;
  st.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count();
}
void Engine::wait_stream() {
  HostSpan w(this, "host_wait_device");
  CPX_HIP(hipStreamSynchronize(stream_));
}

// Batches of at least CPX_DEVICE_MIN_BATCH proofs (default 96: measured cross-over with the spin team of the host-driven path,
// 21.9 against 35.1 ms per prove + verify pass at 32 proofs, 36.3 against 43.6 ms at 96, 59 against 54 ms at 192) run the whole protocol on the GPU (engine_device.cpp).  Smaller
// batches are driven from the host: a lone transcript is latency-bound on a GPU wave (~4 us per Keccak permutation against
// ~0.25 us on a host core, ~1000 permutations per proof), and the host has idle cores.  In the host-driven mode the
// transcript prefix of every loaded proof is hashed on the host as well.
bool Engine::device_prefix(size_t B) const {
  return B >= (size_t)opt_.device_min_batch;
}
// hipStreamSynchronize spins on a host core; an event created with hipEventBlockingSync puts the thread to sleep instead.  With
// one thread per engine context and per rank, and hosts that give a container a small CPU quota, that matters.
void Engine::wait_stream_blocking() {
  if (!ev_block_) CPX_HIP(hipEventCreateWithFlags(&ev_block_, hipEventBlockingSync | hipEventDisableTiming));
  CPX_HIP(hipEventRecord(ev_block_, stream_));
  HostSpan w(this, "host_wait_device");
  CPX_HIP(hipEventSynchronize(ev_block_));
}

void Engine::transcript_prefix_async(size_t B) {
  if (!side_.ev2) CPX_HIP(hipEventCreateWithFlags(&side_.ev2, hipEventDisableTiming));
  CPX_HIP(hipEventRecord(side_.ev2, stream_));
  CPX_HIP(hipStreamWaitEvent(side_.stream, side_.ev2, 0));
  CPX_HIP(hipMemcpyAsync(h_inst_comp_.p, d_bytes_.p, B * 4 * ell_ * 48, hipMemcpyDeviceToHost, side_.stream));
  CPX_HIP(hipMemcpyAsync(h_mcomp_.p, d_mcomp_.p, B * 48, hipMemcpyDeviceToHost, side_.stream));
  CPX_HIP(hipEventRecord(side_.ev, side_.stream));
}
void Engine::wait_side() {
  HostSpan w(this, "host_wait_device");
  CPX_HIP(hipEventSynchronize(side_.ev));
}

const uint32_t* Engine::idx_list(const std::vector<uint32_t>& v) {
  auto it = idx_cache_.find(v);
  if (it != idx_cache_.end()) return it->second;
  uint32_t* d = nullptr;
  CPX_HIP(hipMalloc(&d, v.size() * sizeof(uint32_t)));
  CPX_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  idx_allocs_.push_back(d);
  idx_cache_[v] = d;
  return d;
}

// ---------------------------------------------------------------- generic phases
// Uploads the scalars of all requests, runs accumulate + tails + finalize; affine results are
// scattered to d_pp_[dst]; compressed results (48 B each, request order) returned if comp_out != null.
void Engine::run_msm_phase(const std::vector<MsmReq>& reqs, std::vector<uint8_t>* comp_out) {
  const size_t nt = reqs.size();
  if (!nt) return;
  size_t total = 0;
  for (auto& r : reqs) total += r.n;
  d_scal_.ensure(total);
  d_tasks_.ensure(nt);
  d_wsum_.ensure(nt * 64);
  d_part_.ensure(nt * 8);
  d_res_.ensure(nt);
  d_dst_.ensure(nt);
  d_comp_.ensure(nt * 48);
  const size_t stage_bytes = total * sizeof(Fr) + nt * sizeof(MsmTask) + nt * sizeof(uint32_t);
  h_stage_.ensure(stage_bytes);
  Fr* hs = reinterpret_cast<Fr*>(h_stage_.p);
  MsmTask* ht = reinterpret_cast<MsmTask*>(h_stage_.p + total * sizeof(Fr));
  uint32_t* hd = reinterpret_cast<uint32_t*>(h_stage_.p + total * sizeof(Fr) + nt * sizeof(MsmTask));
  std::vector<size_t> soff(nt);
  size_t off = 0;
  for (size_t i = 0; i < nt; i++) {
    soff[i] = off;
    off += reqs[i].n;
  }
  const double alg = 128.0 * (double)total;   // 96 B affine base + 32 B scalar per MSM point
  parallel_for(nt, [&](size_t i) {
    const MsmReq& r = reqs[i];
    Fr* d = hs + soff[i];
    for (uint32_t j = 0; j < r.n; j++) d[j] = r.scalars[j].f;
    ht[i] = MsmTask{r.bases, r.idx, d_scal_.p + soff[i], r.n, 0, (uint32_t)soff[i]};
    hd[i] = r.dst;
  });
  CPX_HIP(hipMemcpyAsync(d_scal_.p, hs, total * sizeof(Fr), hipMemcpyHostToDevice, stream_));
  CPX_HIP(hipMemcpyAsync(d_tasks_.p, ht, nt * sizeof(MsmTask), hipMemcpyHostToDevice, stream_));
  CPX_HIP(hipMemcpyAsync(d_dst_.p, hd, nt * sizeof(uint32_t), hipMemcpyHostToDevice, stream_));
  uint32_t max_n = 0;
  for (auto& r : reqs) max_n = std::max(max_n, r.n);
  d_conv_.ensure(std::max<size_t>(total, 1));
  tick("k_msm_accw", alg, (double)total);
  launch_msm_accum(d_tasks_.p, (int)nt, (int)max_n, d_conv_.p, d_wsum_.p, stream_);
  tock();
  tick("k_msm_tail", 0, (double)nt * 8);
  launch_msm_tail(opt_, d_wsum_.p, d_part_.p, nullptr, (int)nt * 8, 8, 4, stream_);
  tock();
  tick("k_msm_tail", 0, (double)nt);
  launch_msm_tail(opt_, d_part_.p, nullptr, d_res_.p, (int)nt, 8, 32, stream_);
  tock();
  tick("k_finalize", 0, (double)nt);
  launch_finalize(d_res_.p, (int)nt, d_pp_.p, d_dst_.p, d_comp_.p, stream_);
  tock();
  if (comp_out) {
    h_comp_.ensure(nt * 48);
    CPX_HIP(hipMemcpyAsync(h_comp_.p, d_comp_.p, nt * 48, hipMemcpyDeviceToHost, stream_));
    wait_stream();
    comp_out->assign(h_comp_.p, h_comp_.p + nt * 48);
  }
}

// tasks[i].scalars must already point into d_scal_ (offsets in Fr units from its base are set by the caller
// through `scalars`/`nscalars`, uploaded here).
void Engine::run_smul(const std::vector<SmulTask>& tasks, int cnt, const S* scalars, size_t nscalars, double alg_bytes) {
  if (tasks.empty() || cnt <= 0) return;
  d_stasks_.ensure(tasks.size());
  h_stage_.ensure(nscalars * sizeof(Fr) + tasks.size() * sizeof(SmulTask));
  Fr* hs = reinterpret_cast<Fr*>(h_stage_.p);
  for (size_t i = 0; i < nscalars; i++) hs[i] = scalars[i].f;
  SmulTask* ht = reinterpret_cast<SmulTask*>(h_stage_.p + nscalars * sizeof(Fr));
  memcpy(ht, tasks.data(), tasks.size() * sizeof(SmulTask));
  // the staging buffer may still be in flight from a previous async copy on this stream only if the caller
  // did not synchronise; every phase ends with a synchronising D2H or the explicit sync below.
  CPX_HIP(hipMemcpyAsync(d_scal_.p, hs, nscalars * sizeof(Fr), hipMemcpyHostToDevice, stream_));
  CPX_HIP(hipMemcpyAsync(d_stasks_.p, ht, tasks.size() * sizeof(SmulTask), hipMemcpyHostToDevice, stream_));
  tick("k_smul", alg_bytes, (double)tasks.size() * cnt);
  launch_smul(d_stasks_.p, (int)tasks.size(), cnt, stream_);
  tock();
  wait_stream();   // staging buffer reuse safety
}

// raw lane accumulators of the MSM waves (kernels.h) and their reduction to partial sums in d_tpart_
void Engine::ensure_raw_sets(size_t nsets) {
  d_raw_.ensure(std::max<size_t>(nsets, 1) * raw_set_words());
  d_rawslot_.ensure(std::max<size_t>(nsets, 1));
  d_rawmid_.ensure(std::max<size_t>(nsets, 1) * reduce_mid_per_set());
}
void Engine::reduce_sets(size_t nplain, size_t nweighted, TJac* part) {
  tick("k_reduce_sets", 0, (double)(nplain + nweighted), true);   // up to four launches (groups of 8 lanes, then the groups of a set; plain / bucket sets)
  launch_reduce_sets(opt_, d_raw_.p, d_rawslot_.p, (int)nplain, (int)nweighted, d_rawmid_.p, part ? part : d_tpart_.p, stream_, (int)B_);
  tock();
}

// Table-backed MSM phase: the CRS segments of a request go to k_msm_fix, the per-proof segments to k_msm_tblw (one
// wave per task and window group); k_reduce_sets turns the raw lane accumulators into partial sums, k_finalize_ranges
// adds them per request, normalises and compresses.
void Engine::run_tbl_phase(const std::vector<TblReq>& reqs, std::vector<uint8_t>* comp_out) {
  const size_t nt = reqs.size();
  if (!nt) return;
  const bool fix = fix_bits_ && fixtab();
  const TAff* c_lo = ctab();
  const TAff* c_hi = ctab() + (size_t)copies_ * nc();
  auto is_crs = [&](const TblSeg& sg) { return fix && sg.n && sg.base >= c_lo && sg.base < c_hi; };
  auto needs_tbl = [&](const TblReq& r) { return !fix || (r.seg0.n && !is_crs(r.seg0)) || (r.seg1.n && !is_crs(r.seg1)); };
  size_t ntt_total = 0, nft_total = 0;
  for (const TblReq& r : reqs) {
    ntt_total += needs_tbl(r) ? 1 : 0;
    nft_total += (is_crs(r.seg0) ? 1 : 0) + (is_crs(r.seg1) ? 1 : 0);
  }
  const int fix_wpw = fix ? msm_fix_windows_per_wave(opt_, (int)nft_total, fix_bits_) : 16;
  const uint32_t fix_parts = fix ? (uint32_t)msm_fix_parts(fix_bits_, fix_wpw) : 0;
  const int tbl_wpw = msm_tblw_windows_per_wave(opt_, (int)ntt_total);   // windows per wave of the shifted-table kernel
  uint32_t tbl_max_n = 0;
  for (const TblReq& r : reqs)
    if (needs_tbl(r)) tbl_max_n = std::max(tbl_max_n, (is_crs(r.seg0) ? 0u : r.seg0.n) + (is_crs(r.seg1) ? 0u : r.seg1.n));
  const int tbl_slices = msm_tblw_slices(opt_, (int)ntt_total, tbl_wpw, (int)tbl_max_n);   // a lone proof: several waves share a task's points
  const uint32_t tbl_parts = (uint32_t)(msm_tblw_parts(tbl_wpw) * tbl_slices);   // partial sums (2 per wave) per task
  // layout: per request its scalars (seg0 then seg1); shifted-table tasks and fixed-base tasks; partial ranges
  std::vector<size_t> soff(nt);
  std::vector<uint32_t> pfirst(nt), pcount(nt), tt_index(nt, ~0u), ft_index(nt * 2, ~0u);
  size_t total = 0, ntt = 0, nft = 0, nparts = 0;
  double pts_tbl = 0, pts_fix = 0;
  for (size_t i = 0; i < nt; i++) {
    const TblReq& r = reqs[i];
    soff[i] = total;
    if (!r.dev) total += r.seg0.n + r.seg1.n;
    pfirst[i] = (uint32_t)nparts;
    const bool f0 = is_crs(r.seg0), f1 = is_crs(r.seg1);
    if (needs_tbl(r)) {
      tt_index[i] = (uint32_t)ntt++;
      nparts += tbl_parts;
    }
    if (f0) {
      ft_index[2 * i] = (uint32_t)nft++;
      nparts += fix_parts;
    }
    if (f1) {
      ft_index[2 * i + 1] = (uint32_t)nft++;
      nparts += fix_parts;
    }
    pcount[i] = (uint32_t)nparts - pfirst[i];
    pts_fix += (f0 ? r.seg0.n : 0) + (f1 ? r.seg1.n : 0);
    pts_tbl += (f0 ? 0 : r.seg0.n) + (f1 ? 0 : r.seg1.n);
  }
  d_tpart_.ensure(nparts);
  d_comp_.ensure(nt * 48);
  // everything the phase's kernels read — scalars, task descriptors, partial ranges, destinations, addends — is staged in ONE pinned
  // buffer with the layout [Fr scalars | TblTask | FixTask | u32 arrays] and uploaded with ONE copy into a device blob of the same
  // layout (a handful of separate small copies cost ~30 us of host time per phase, which a lone proof waits for)
  const size_t b_scal = total * sizeof(Fr), b_tt = ntt * sizeof(TblTask), b_ft = nft * sizeof(FixTask);
  static_assert(sizeof(TblTask) % 8 == 0 && sizeof(FixTask) % 8 == 0 && sizeof(Fr) % 8 == 0, "blob sections keep pointer alignment");
  bool any_add = false;
  for (const TblReq& r : reqs) any_add |= r.add[0] != ~0u;
  const size_t b_blob = b_scal + b_tt + b_ft + 6 * nt * sizeof(uint32_t);
  h_stage_.ensure(b_blob);
  d_blob_.ensure(b_blob);
  Fr* const d_scal = reinterpret_cast<Fr*>(d_blob_.p);
  TblTask* const d_tt = reinterpret_cast<TblTask*>(d_blob_.p + b_scal);
  FixTask* const d_ft = reinterpret_cast<FixTask*>(d_blob_.p + b_scal + b_tt);
  uint32_t* const d_u32 = reinterpret_cast<uint32_t*>(d_blob_.p + b_scal + b_tt + b_ft);
  const uint32_t dummy_dst = slot_index(0, SlotMap(L_).TMP(7));   // results nobody reads land in a scratch slot
  Fr* hs = reinterpret_cast<Fr*>(h_stage_.p);
  TblTask* ht = reinterpret_cast<TblTask*>(h_stage_.p + b_scal);
  FixTask* hf = reinterpret_cast<FixTask*>(h_stage_.p + b_scal + b_tt);
  uint32_t* hp = reinterpret_cast<uint32_t*>(h_stage_.p + b_scal + b_tt + b_ft);
  const TblSeg none{nullptr, nullptr, 0, 0};
  parallel_for(nt, [&](size_t i) {
    const TblReq& r = reqs[i];
    const Fr* sbase = r.dev ? r.dev : d_scal + soff[i];   // where this request's scalars are (or will be) on the device
    if (!r.dev) {
      Fr* d = hs + soff[i];
      for (uint32_t j = 0; j < r.seg0.n; j++) d[j] = r.s0[j].f;
      for (uint32_t j = 0; j < r.seg1.n; j++) d[r.seg0.n + j] = r.s1[j].f;
    }
    const bool f0 = is_crs(r.seg0), f1 = is_crs(r.seg1);
    uint32_t slot = pfirst[i];
    if (tt_index[i] != ~0u) {
      TblTask t;
      // the shifted-table task keeps the scalar layout of the request; fixed-base segments are emptied
      t.seg[0] = f0 ? TblSeg{r.seg0.base, r.seg0.idx, r.seg0.copy_stride, 0} : r.seg0;
      t.seg[1] = f1 ? none : r.seg1;
      t.scalars = sbase + (f0 ? r.seg0.n : 0);
      if (f0) {   // seg0 removed: its scalars are skipped, seg1 becomes the only segment
        t.seg[0] = f1 ? none : r.seg1;
        t.seg[1] = none;
      }
      t.flags = 0;
      t.pad = slot;
      t.digits = nullptr;
      ht[tt_index[i]] = t;
      slot += tbl_parts;
    }
    if (f0) {
      hf[ft_index[2 * i]] = FixTask{r.seg0.idx, sbase, (uint32_t)(r.seg0.base - c_lo), r.seg0.n, 0, slot};
      slot += fix_parts;
    }
    if (f1) {
      hf[ft_index[2 * i + 1]] = FixTask{r.seg1.idx, sbase + r.seg0.n, (uint32_t)(r.seg1.base - c_lo), r.seg1.n, 0, slot};
      slot += fix_parts;
    }
    hp[i] = pfirst[i];
    hp[nt + i] = pcount[i];
    hp[2 * nt + i] = r.dst != ~0u ? r.dst : dummy_dst;
    for (int j = 0; j < 3; j++) hp[3 * nt + 3 * i + j] = r.add[j];
  });
  CPX_HIP(hipMemcpyAsync(d_blob_.p, h_stage_.p, b_blob, hipMemcpyHostToDevice, stream_));
  const size_t fix_sets = nft * fix_parts, tbl_sets = ntt * tbl_parts;   // raw sets: [fixed-base waves | bucket sets of the table waves]
  ensure_raw_sets(fix_sets + tbl_sets);
  if (nft && ntt && fix_bits_ == 16 && fix_wpw == 2 && tbl_wpw == 2) {   // a lone proof: both MSM kernels of the phase in one launch
    tick("k_msm_fix_tblw", 128.0 * (pts_fix + pts_tbl), pts_fix + pts_tbl);
    launch_msm_fix_tblw(d_ft, (int)nft, fixtab(), (int)nc(), d_raw_.p, d_rawslot_.p, d_tt, (int)ntt, tbl_slices, d_raw_.p + fix_sets * raw_set_words(),
                        d_rawslot_.p + fix_sets, stream_);
    tock();
  } else {
  if (nft) {
    tick(fix_kernel_name(fix_bits_, fix_wpw), 128.0 * pts_fix, pts_fix);
    launch_msm_fix(d_ft, (int)nft, fixtab(), fix_bits_, fix_wpw, (int)nc(), d_raw_.p, d_rawslot_.p, stream_);
    tock();
  }
  if (ntt) {
    tick(tblw_kernel_name(tbl_wpw), 128.0 * pts_tbl, pts_tbl);
    launch_msm_tblw(d_tt, (int)ntt, tbl_wpw, d_raw_.p + fix_sets * raw_set_words(), d_rawslot_.p + fix_sets, stream_, tbl_slices);
    tock();
  }
  }
  reduce_sets(fix_sets, tbl_sets);
  tick("k_finalize_ranges", 0, (double)nt);
  launch_finalize_ranges(opt_, d_tpart_.p, d_u32, d_u32 + nt, (int)nt, d_pp_.p, d_u32 + 2 * nt, d_comp_.p, stream_, any_add ? d_u32 + 3 * nt : nullptr);
  tock();
  if (comp_out) {
    h_comp_.ensure(nt * 48);
    CPX_HIP(hipMemcpyAsync(h_comp_.p, d_comp_.p, nt * 48, hipMemcpyDeviceToHost, stream_));
    wait_stream();
    comp_out->assign(h_comp_.p, h_comp_.p + nt * 48);
  }
}

// ---------------------------------------------------------------- CRS
void Engine::set_crs(size_t ell, const uint8_t* points) {
  const size_t n = ell + N_BLINDERS;
  if (ell == 0 || (n & (n - 1))) throw std::invalid_argument("ell + 4 must be a power of two");
  // the cached device plans were laid out for the previous CRS (they embed its compressed H and table addresses, which a new
  // allocation of the same size may reuse): a new CRS — also a refused one — starts from no plans
  dprove_.signature.clear();
  dverify_.signature.clear();
  try {
    set_crs_impl(ell, points);
  } catch (...) {   // e.g. out of device memory while building the tables: the context is left without a CRS, not half-initialised
    ell_ = n_ = L_ = 0;
    B_ = 0;
    consts_rows_ = 0;
    crs_tab_.reset();
    throw;
  }
}
void Engine::set_crs_impl(size_t ell, const uint8_t* points) {
  const size_t n = ell + N_BLINDERS;
  CPX_HIP(hipSetDevice(device_));
  ell_ = ell;
  n_ = n;
  L_ = 0;
  while ((size_t(1) << L_) < n) L_++;
  const Aff* pts = reinterpret_cast<const Aff*>(points);
  crs_host_.assign(pts, pts + ell + 7);
  std::vector<Aff> crs(n + 1), gb(n);
  for (size_t i = 0; i < n; i++) crs[i] = pts[i];   // G | Hvec
  crs[n] = pts[n];                                  // H
  for (size_t i = 0; i < ell + 2; i++) gb[i] = pts[i];
  gb[ell + 2] = pts[n + 1];   // G_t
  gb[ell + 3] = pts[n + 2];   // G_u
  d_crs_.ensure(n + 1);
  d_crs_gb_.ensure(n);
  CPX_HIP(hipMemcpy(d_crs_.p, crs.data(), (n + 1) * sizeof(Aff), hipMemcpyHostToDevice));
  CPX_HIP(hipMemcpy(d_crs_gb_.p, gb.data(), n * sizeof(Aff), hipMemcpyHostToDevice));
  crs_single_[0] = pts[n];
  crs_single_[1] = pts[n + 1];
  crs_single_[2] = pts[n + 2];
  // G_sum, H_sum (crs.rs:46-47) as unit-scalar MSMs on the device
  B_ = 0;
  consts_rows_ = 0;
  pp_stride_ = 4 * ell_ + SlotMap(L_).count();
  d_pp_.ensure(pp_stride_);
  SVec ones(ell, S::one());
  std::vector<MsmReq> reqs;
  reqs.push_back(MsmReq{d_crs_.p, nullptr, ones.data(), (uint32_t)ell, 0});
  reqs.push_back(MsmReq{d_crs_.p + ell, nullptr, ones.data(), (uint32_t)N_BLINDERS, 1});
  std::vector<uint8_t> comp;
  run_msm_phase(reqs, &comp);
  Aff sums[2];
  CPX_HIP(hipMemcpy(sums, d_pp_.p, 2 * sizeof(Aff), hipMemcpyDeviceToHost));
  crs_single_[3] = sums[0];
  crs_single_[4] = sums[1];
  // compressed H (needed for the vec_T/vec_U blinder slots in the transcript)
  d_comp_.ensure(48);
  CPX_HIP(hipMemcpy(d_pp_.p, &crs_single_[0], sizeof(Aff), hipMemcpyHostToDevice));
  launch_compress(d_pp_.p, 1, 1, 1, d_comp_.p, stream_);
  CPX_HIP(hipMemcpyAsync(crs_H_comp_, d_comp_.p, 48, hipMemcpyDeviceToHost, stream_));
  CPX_HIP(hipStreamSynchronize(stream_));
  // shifted-base table + fixed-base table of multiples: built once per (device, CRS), shared by every proof and
  // by every engine of this process on the device
  {
    static std::mutex reg_mu;
    static std::vector<std::weak_ptr<CrsTables>> registry;
    std::lock_guard<std::mutex> lk(reg_mu);
    const size_t NC = nc();
    int want_fix = fix_bits_cfg_;
    // 19 bits: 14 windows x 2^18 multiples = 122 GB at ell = 252 (leave room for the batches: at most 45 % of the HBM); 16 bits: 17.5 GB
    // at ell = 252, 70 GB at ell = 1020; fall back 19 -> 16 -> 8 (0.1 GB / 0.4 GB) when HBM is short
    for (int cand : {19, 16}) {
      if (want_fix != cand) continue;
      size_t free_b = 0, total_b = 0;
      CPX_HIP(hipMemGetInfo(&free_b, &total_b));
      const size_t Wc = (size_t)msm_fix_windows(cand), segs = cand == 19 ? 8 : 1;
      const size_t need = Wc * ((size_t)1 << (cand - 1)) * NC * sizeof(TFix) + (Wc * NC * segs + 63) * 256 * sizeof(TblTmp);
      bool have = false;   // an existing shared table costs nothing
      for (auto& w : registry)
        if (auto sp = w.lock()) have |= sp->device == device_ && sp->fix_bits == cand && sp->key.size() == (ell + 7) * sizeof(Aff) && !memcmp(sp->key.data(), points, sp->key.size());
      if (!have && (need + (need >> 3) > free_b || (cand == 19 && need > total_b / 20 * 9))) want_fix = cand == 19 ? 16 : 8;
    }
    fix_bits_ = want_fix;
    std::vector<uint8_t> key(points, points + (ell + 7) * sizeof(Aff));
    crs_tab_.reset();
    for (auto it = registry.begin(); it != registry.end();) {
      auto sp = it->lock();
      if (!sp) {
        it = registry.erase(it);
        continue;
      }
      if (sp->device == device_ && sp->fix_bits == want_fix && sp->key == key) crs_tab_ = sp;
      ++it;
    }
    if (!crs_tab_) {
      auto tab = std::make_shared<CrsTables>();
      tab->device = device_;
      tab->fix_bits = want_fix;
      tab->key = std::move(key);
      tab->ctab.ensure((size_t)copies_ * NC);
      std::vector<Aff> row(NC);
      for (size_t i = 0; i < n + 3; i++) row[i] = pts[i];   // G | Hvec | H | G_t | G_u
      row[n + 3] = crs_single_[3];                          // G_sum, H_sum (crs.rs:46-47): B, D are built from them
      row[n + 4] = crs_single_[4];
      DevBuf<Aff> d_row;
      d_row.ensure(NC);
      CPX_HIP(hipMemcpy(d_row.p, row.data(), NC * sizeof(Aff), hipMemcpyHostToDevice));
      DevBuf<TblTmp> tmp;   // build scratch, released afterwards
      tmp.ensure(NC * (size_t)(copies_ - 1));
      launch_table_build(opt_, d_row.p, 0, tab->ctab.p, 1, 0, (int)NC, (int)NC, copies_, true, tmp.p, stream_);
      CPX_HIP(hipStreamSynchronize(stream_));
      if (want_fix) {
        // multiples m * 2^(c w) * P, m <= 2^(c-1): shifted copies first, then the multiples
        const int W = msm_fix_windows(want_fix);
        const size_t M = size_t(1) << (want_fix - 1);
        DevBuf<TAff> d_shift;
        d_shift.ensure((size_t)W * NC);
        launch_table_build(opt_, d_row.p, 0, d_shift.p, 1, 0, (int)NC, (int)NC, W, false, tmp.p, stream_, want_fix);   // copy w = 2^(want_fix w) P
        tab->fixtab.ensure((size_t)W * M * NC);
        const int chunk = (int)std::min<size_t>(256, M);
        const int segs = want_fix == 19 ? 8 : 1;   // 2^18 multiples per (window, base): eight threads of 2^15 each
        const size_t threads = ((size_t)W * NC * segs + 63) / 64 * 64;
        tmp.ensure(threads * chunk);
        launch_fix_build(d_shift.p, (int)NC, want_fix, tab->fixtab.p, tmp.p, chunk, stream_, segs);
        CPX_HIP(hipStreamSynchronize(stream_));
      }
      registry.push_back(tab);
      crs_tab_ = tab;
    }
  }
}
void Engine::crs_sums(uint8_t* g_sum, uint8_t* h_sum) const {
  memcpy(g_sum, &crs_single_[3], sizeof(Aff));
  memcpy(h_sum, &crs_single_[4], sizeof(Aff));
}

// ---------------------------------------------------------------- tier 0
void Engine::msm(const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t* out_jac) {
  CPX_HIP(hipSetDevice(device_));
  DevBuf<Aff>& db = t0_.a0;   // (tier-0 scratch of the engine, engine.hpp: no allocation on the call path once warm)
  DevBuf<Fr>& ds = t0_.fr;
  DevBuf<MsmTask>& dt = t0_.mtask;
  DevBuf<TJac>&w = t0_.w, &pt = t0_.pt;
  DevBuf<Jac>& res = t0_.res;
  struct Trim {
    Engine* e;
    ~Trim() {
      tier0_trim(e->t0_.a0);
      tier0_trim(e->t0_.fr);
      tier0_trim(e->t0_.conv);
      tier0_trim(e->t0_.dig);
    }
  } trim{this};
  db.ensure(std::max<size_t>(n, 1));
  ds.ensure(std::max<size_t>(n, 1));
  dt.ensure(1);
  w.ensure(64);
  pt.ensure(8);
  res.ensure(1);
  if (n) {
    CPX_HIP(hipMemcpyAsync(db.p, bases, n * sizeof(Aff), hipMemcpyHostToDevice, stream_));
    CPX_HIP(hipMemcpyAsync(ds.p, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, stream_));
  }
  MsmTask t{db.p, nullptr, ds.p, (uint32_t)n, 0, 0};
  CPX_HIP(hipMemcpyAsync(dt.p, &t, sizeof t, hipMemcpyHostToDevice, stream_));
  DevBuf<TAff>& conv = t0_.conv;
  conv.ensure(2 * std::max<size_t>(n, 1));
  if ((long)n >= opt_.msm_endo_min && n) {   // endomorphism split + radix-256 bucket lists (the verifier's kernel): 32 additions per point
    DevBuf<TblTask>& tt = t0_.ttask;
    DevBuf<TJac>& part = t0_.part;
    DevBuf<uint32_t>& dig = t0_.dig;
    const int slices = msm_tblw_slices(opt_, 1, 2, (int)n);
    tt.ensure(1);
    part.ensure(32 * (size_t)slices);
    dig.ensure(9 * n);
    ensure_raw_sets(32 * (size_t)slices);
    tick("k_msm_tblw<2, true>", 128.0 * n, (double)n);
    launch_msm_endo(dt.p, 1, (int)n, conv.p, dig.p, tt.p, d_raw_.p, d_rawslot_.p, stream_, slices);
    tock();
    reduce_sets(0, 32 * (size_t)slices, part.p);
    launch_msm_tail(opt_, part.p, nullptr, res.p, 1, 16, 8, stream_, nullptr, 0, 2 * slices);
    CPX_HIP(hipMemcpyAsync(out_jac, res.p, sizeof(Jac), hipMemcpyDeviceToHost, stream_));
    CPX_HIP(hipStreamSynchronize(stream_));   // the result is in out_jac; the scratch may be reused by the next call
    flush_timers();
    return;
  }
  tick("k_msm_accw", 128.0 * n, (double)n);
  launch_msm_accum(dt.p, 1, (int)n, conv.p, w.p, stream_);
  tock();
  launch_msm_tail(opt_, w.p, pt.p, nullptr, 8, 8, 4, stream_);
  launch_msm_tail(opt_, pt.p, nullptr, res.p, 1, 8, 32, stream_);
  CPX_HIP(hipMemcpyAsync(out_jac, res.p, sizeof(Jac), hipMemcpyDeviceToHost, stream_));
  CPX_HIP(hipStreamSynchronize(stream_));
  flush_timers();
}
void Engine::normalize(const uint8_t* jac, size_t n, uint8_t* out_aff, uint8_t* out_comp) {
  CPX_HIP(hipSetDevice(device_));
  if (!n) return;
  DevBuf<Jac>& dj = t0_.jin;
  DevBuf<Aff>& da = t0_.a0;
  DevBuf<uint8_t>& dc = t0_.bytes;
  dj.ensure(n);
  da.ensure(n);
  dc.ensure(n * 48);
  CPX_HIP(hipMemcpyAsync(dj.p, jac, n * sizeof(Jac), hipMemcpyHostToDevice, stream_));
  launch_finalize(dj.p, (int)n, da.p, nullptr, dc.p, stream_);
  if (out_aff) CPX_HIP(hipMemcpyAsync(out_aff, da.p, n * sizeof(Aff), hipMemcpyDeviceToHost, stream_));
  if (out_comp) CPX_HIP(hipMemcpyAsync(out_comp, dc.p, n * 48, hipMemcpyDeviceToHost, stream_));
  CPX_HIP(hipStreamSynchronize(stream_));
  tier0_trim(dj);
  tier0_trim(da);
  tier0_trim(dc);
}
bool Engine::sum_jac(const uint8_t* points_jac, size_t n, uint8_t* out_jac) {
  CPX_HIP(hipSetDevice(device_));
  DevBuf<Jac>&din = t0_.jin, &dout = t0_.res;
  DevBuf<int>& dflag = t0_.flag;
  din.ensure(std::max<size_t>(n, 1));
  dout.ensure(1);
  dflag.ensure(1);
  if (n) CPX_HIP(hipMemcpyAsync(din.p, points_jac, n * sizeof(Jac), hipMemcpyHostToDevice, stream_));
  launch_sum_jac(din.p, (int)n, dout.p, dflag.p, stream_);
  int flag = 0;
  CPX_HIP(hipMemcpyAsync(out_jac, dout.p, sizeof(Jac), hipMemcpyDeviceToHost, stream_));
  CPX_HIP(hipMemcpyAsync(&flag, dflag.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
  CPX_HIP(hipStreamSynchronize(stream_));
  tier0_trim(din);
  return flag != 0;
}
void Engine::msm_jac(const uint8_t* bases_jac, const uint8_t* scalars, size_t n, uint8_t* out_jac) {
  std::vector<uint8_t> aff(std::max<size_t>(n, 1) * sizeof(Aff));
  normalize(bases_jac, n, aff.data(), nullptr);
  msm(aff.data(), scalars, n, out_jac);
}
void Engine::fold(uint8_t* PL, const uint8_t* PR, const uint8_t* gamma, size_t half) {
  CPX_HIP(hipSetDevice(device_));
  if (!half) return;
  DevBuf<Aff>&dl = t0_.a0, &dr = t0_.a1;
  DevBuf<Fr>& dg = t0_.fr;
  DevBuf<SmulTask>& dt = t0_.stask;
  dl.ensure(half);
  dr.ensure(half);
  dg.ensure(1);
  dt.ensure(1);
  CPX_HIP(hipMemcpyAsync(dl.p, PL, half * sizeof(Aff), hipMemcpyHostToDevice, stream_));
  CPX_HIP(hipMemcpyAsync(dr.p, PR, half * sizeof(Aff), hipMemcpyHostToDevice, stream_));
  CPX_HIP(hipMemcpyAsync(dg.p, gamma, sizeof(Fr), hipMemcpyHostToDevice, stream_));
  SmulTask t{dl.p, dr.p, dl.p, dg.p, 0, opt_.scale_any_point ? SMUL_PLAIN : 0u};
  CPX_HIP(hipMemcpyAsync(dt.p, &t, sizeof t, hipMemcpyHostToDevice, stream_));
  tick("k_smul", 288.0 * half, (double)half);
  launch_smul(dt.p, 1, (int)half, stream_);
  tock();
  CPX_HIP(hipMemcpyAsync(PL, dl.p, half * sizeof(Aff), hipMemcpyDeviceToHost, stream_));
  CPX_HIP(hipStreamSynchronize(stream_));
  flush_timers();
  tier0_trim(dl);
  tier0_trim(dr);
}
void Engine::scale(const uint8_t* P, const uint8_t* scalars, size_t scalar_stride, size_t n, uint8_t* out) {
  CPX_HIP(hipSetDevice(device_));
  if (!n) return;
  DevBuf<Aff>&dp = t0_.a0, &dout = t0_.a1;
  DevBuf<Fr>& dsc = t0_.fr;
  DevBuf<SmulTask>& dt = t0_.stask;
  const size_t ns = scalar_stride ? n : 1;
  dp.ensure(n);
  dout.ensure(n);
  dsc.ensure(ns);
  dt.ensure(1);
  CPX_HIP(hipMemcpyAsync(dp.p, P, n * sizeof(Aff), hipMemcpyHostToDevice, stream_));
  CPX_HIP(hipMemcpyAsync(dsc.p, scalars, ns * sizeof(Fr), hipMemcpyHostToDevice, stream_));
  SmulTask t{nullptr, dp.p, dout.p, dsc.p, scalar_stride ? 1u : 0u, opt_.scale_any_point ? SMUL_PLAIN : 0u};
  CPX_HIP(hipMemcpyAsync(dt.p, &t, sizeof t, hipMemcpyHostToDevice, stream_));
  tick("k_smul", 224.0 * n, (double)n);
  launch_smul(dt.p, 1, (int)n, stream_);
  tock();
  CPX_HIP(hipMemcpyAsync(out, dout.p, n * sizeof(Aff), hipMemcpyDeviceToHost, stream_));
  CPX_HIP(hipStreamSynchronize(stream_));
  flush_timers();
  tier0_trim(dp);
  tier0_trim(dout);
  tier0_trim(dsc);
}
// Encodings with the infinity flag set (include/cpx.h, option strict_infinity).  ark-bls12-381 ^0.4's `read_g1_compressed` — the
// deserialiser behind `G1Affine::deserialize_compressed`, whisk.rs:313-320 — returns the identity as soon as the compression and the
// infinity flag are set, without looking at the sort flag or the other bits (recalled from the 0.4.0 source; 0.5 added both checks), and the
// verifier then hashes the point's CANONICAL serialisation (transcript.rs:28-36 append the deserialised points).  The device kernels
// accept exactly the canonical form, so with strict_infinity = 0 the host rewrites a non-canonical infinity encoding to 0xc0 || 0^47 in a
// copy of the input before anything reads it; with strict_infinity = 1 the bytes go through untouched and such an encoding is a
// deserialisation error.  Returns true if `enc` was rewritten.
static inline bool infinity_is_noncanonical(const uint8_t* enc) {
  if ((enc[0] & 0xc0) != 0xc0) return false;
  if (enc[0] != 0xc0) return true;
  for (int i = 1; i < 48; i++)
    if (enc[i]) return true;
  return false;
}
const uint8_t* Engine::canonical_infinities(const uint8_t* bytes, size_t nbytes, size_t nrec, size_t rec_stride, const std::vector<size_t>& offsets) {
  if (opt_.strict_infinity) return bytes;
  bool copied = false;
  for (size_t r = 0; r < nrec; r++)
    for (size_t off : offsets) {
      const size_t at = r * rec_stride + off;
      if (!(bytes[at] & 0x40) || !infinity_is_noncanonical(bytes + at)) continue;   // (one byte per encoding on the common path)
      if (!copied) {
        canon_bytes_.assign(bytes, bytes + nbytes);
        bytes = canon_bytes_.data();
        copied = true;
      }
      canon_bytes_[at] = 0xc0;
      memset(&canon_bytes_[at + 1], 0, 47);
    }
  return bytes;
}

int Engine::decompress(const uint8_t* comp, size_t n, uint8_t* out_aff, int check_subgroup, uint8_t* status_out) {
  CPX_HIP(hipSetDevice(device_));
  if (!n) return CPX_OK;
  comp = canonical_infinities(comp, n * 48, n, 48, {0});
  DevBuf<uint8_t>&dc = t0_.bytes, &dst = t0_.status;
  DevBuf<Aff>& da = t0_.a0;
  dc.ensure(n * 48);
  dst.ensure(n);
  da.ensure(n);
  CPX_HIP(hipMemcpyAsync(dc.p, comp, n * 48, hipMemcpyHostToDevice, stream_));
  launch_decompress(opt_, dc.p, (int)n, da.p, nullptr, dst.p, check_subgroup, stream_);
  std::vector<uint8_t> st(n);
  CPX_HIP(hipMemcpyAsync(out_aff, da.p, n * sizeof(Aff), hipMemcpyDeviceToHost, stream_));
  CPX_HIP(hipMemcpyAsync(st.data(), dst.p, n, hipMemcpyDeviceToHost, stream_));
  CPX_HIP(hipStreamSynchronize(stream_));
  if (status_out) {   // per-point verdicts: the call itself succeeds
    memcpy(status_out, st.data(), n);
    return CPX_OK;
  }
  for (auto s : st)
    if (s) return CPX_ERR_DESERIALIZE;
  return CPX_OK;
}

double Engine::bench_fpmul(int blocks, int iters, int reps) {
  CPX_HIP(hipSetDevice(device_));
  const size_t nth = (size_t)blocks * 256;
  DevBuf<Fp> d;
  d.ensure(2 * nth);
  std::vector<Fp> h(2 * nth);
  for (size_t i = 0; i < 2 * nth; i++) {
    h[i] = Fp::one();
    h[i].v[0] ^= (uint32_t)(i * 2654435761u);
    h[i].v[5] ^= (uint32_t)(i * 40503u);
    h[i].v[11] &= 0x0fffffffu;
  }
  CPX_HIP(hipMemcpy(d.p, h.data(), 2 * nth * sizeof(Fp), hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CPX_HIP(hipEventCreate(&a));
  CPX_HIP(hipEventCreate(&b));
  auto launch = opt_.bench_field == 28 ? launch_bench_f28mul : launch_bench_fpmul;
  launch(d.p, blocks, iters, stream_);   // warm-up
  CPX_HIP(hipEventRecord(a, stream_));
  for (int r = 0; r < reps; r++) launch(d.p, blocks, iters, stream_);
  CPX_HIP(hipEventRecord(b, stream_));
  CPX_HIP(hipStreamSynchronize(stream_));
  float ms = 0;
  CPX_HIP(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return (double)nth * 2.0 * iters * reps / (ms * 1e-3);
}

// ---------------------------------------------------------------- batch load
// The public instance of `batch` proofs -> the engine's instance buffers, from host memory (batch_load) or from the device staging area of
// batch_load_begin (batch_load_end: device-to-device, a few hundred microseconds per GB).  Everything is enqueued on the main stream.
void Engine::load_rows(size_t batch, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T, const uint8_t* vec_U, const uint8_t* M, bool from_device) {
  if (!ell_) throw std::logic_error("set_crs first");
  CPX_HIP(hipSetDevice(device_));
  const hipMemcpyKind kind = from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const SlotMap sm(L_);
  pp_stride_ = 4 * ell_ + sm.count();
  const Aff* pp_before = d_pp_.p;
  const Aff* psrc_before = d_psrc_.p;
  d_pp_.ensure(batch * pp_stride_);
  d_Mjac_.ensure(batch);
  B_ = batch;
  const size_t vb = ell_ * sizeof(Aff);
  // strided 2-D copies: one per instance vector instead of one per proof
  const size_t pitch = pp_stride_ * sizeof(Aff);
  auto rows = [&](Aff* dst, size_t dpitch, const uint8_t* src, size_t spitch, size_t width, hipMemcpyKind k) {
    CPX_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, batch, k, stream_));
  };
  rows(pp(0), pitch, vec_R, vb, vb, kind);
  rows(pp(0) + ell_, pitch, vec_S, vb, vb, kind);
  rows(pp(0) + 2 * ell_, pitch, vec_T, vb, vb, kind);
  rows(pp(0) + 3 * ell_, pitch, vec_U, vb, vb, kind);
  CPX_HIP(hipMemcpyAsync(d_Mjac_.p, M, batch * sizeof(Jac), kind, stream_));
  // copy 0 of the per-proof tables: M (filled at prove time) | T || O O H O | U || O O O H  (curdleproofs.rs:141-155)
  const size_t NP = np();
  d_ptab_.ensure(batch * (size_t)copies_ * NP);
  d_psrc_.ensure(batch * NP);
  // build scratch: the 15 doubled copies of every point — of one chunk of the device prover's table build (engine_device.cpp); the host-driven
  // prover, which builds all rows at once, makes sure of its own (batch_prove_tables)
  d_tbltmp_.ensure(table_chunk_rows(batch) * NP * (size_t)(copies_ / 2 - 1));
  const size_t spitch = NP * sizeof(Aff);
  Aff* t = d_psrc_.p;
  rows(t + 1, spitch, vec_T, vb, vb, kind);
  rows(t + 1 + n_, spitch, vec_U, vb, vb, kind);
  // The constants every proof's row carries (the CRS's single points in five slots; the blinder tails O O H O / O O O H behind T and U) are the
  // same for every batch of this CRS: written when the buffers are new, have grown, or the CRS has changed — not on every load
  if (consts_rows_ < batch || pp_before != d_pp_.p || psrc_before != d_psrc_.p) {
    std::vector<Aff> rep(batch * 5);   // (a pitch of 0 is not a valid 2-D copy)
    for (size_t p = 0; p < batch; p++) memcpy(&rep[5 * p], crs_single_, 5 * sizeof(Aff));
    rows(slot(0, SL_H), pitch, reinterpret_cast<const uint8_t*>(rep.data()), 5 * sizeof(Aff), 5 * sizeof(Aff), hipMemcpyHostToDevice);
    std::vector<Aff> tails(batch * 8, Aff::identity());
    for (size_t p = 0; p < batch; p++) tails[8 * p + 2] = tails[8 * p + 7] = crs_single_[0];
    rows(t + 1 + ell_, spitch, reinterpret_cast<const uint8_t*>(tails.data()), 8 * sizeof(Aff), 4 * sizeof(Aff), hipMemcpyHostToDevice);
    rows(t + 1 + n_ + ell_, spitch, reinterpret_cast<const uint8_t*>(tails.data() + 4), 8 * sizeof(Aff), 4 * sizeof(Aff), hipMemcpyHostToDevice);
    CPX_HIP(hipStreamSynchronize(stream_));   // rep / tails leave scope
    consts_rows_ = batch;
  }
}
void Engine::batch_load(size_t batch, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T, const uint8_t* vec_U, const uint8_t* M) {
  load_rows(batch, vec_R, vec_S, vec_T, vec_U, M, false);
  CPX_HIP(hipStreamSynchronize(stream_));   // the caller's buffers are free again
}
// cpx_batch_load_begin / _end (include/cpx.h): the NEXT batch's instance crosses PCIe on an upload stream of its own while the loaded batch is
// being proven and verified; _end makes it the loaded batch with device-to-device copies behind the main stream's last kernel.
void Engine::batch_load_begin(size_t batch, const uint8_t* vec_R, const uint8_t* vec_S, const uint8_t* vec_T, const uint8_t* vec_U, const uint8_t* M) {
  if (!ell_) throw std::logic_error("set_crs first");
  CPX_HIP(hipSetDevice(device_));
  if (!stage_.stream) {
    CPX_HIP(hipStreamCreateWithFlags(&stage_.stream, hipStreamNonBlocking));
    CPX_HIP(hipEventCreateWithFlags(&stage_.uploaded, hipEventDisableTiming));
    CPX_HIP(hipEventCreateWithFlags(&stage_.consumed, hipEventDisableTiming));
  }
  const size_t pts = batch * ell_;
  for (auto* b : {&stage_.R, &stage_.S, &stage_.T, &stage_.U}) b->ensure(pts);
  stage_.M.ensure(batch);
  if (stage_.have_consumed) CPX_HIP(hipStreamWaitEvent(stage_.stream, stage_.consumed, 0));   // the previous staged batch has been copied out
  CPX_HIP(hipMemcpyAsync(stage_.R.p, vec_R, pts * sizeof(Aff), hipMemcpyHostToDevice, stage_.stream));
  CPX_HIP(hipMemcpyAsync(stage_.S.p, vec_S, pts * sizeof(Aff), hipMemcpyHostToDevice, stage_.stream));
  CPX_HIP(hipMemcpyAsync(stage_.T.p, vec_T, pts * sizeof(Aff), hipMemcpyHostToDevice, stage_.stream));
  CPX_HIP(hipMemcpyAsync(stage_.U.p, vec_U, pts * sizeof(Aff), hipMemcpyHostToDevice, stage_.stream));
  CPX_HIP(hipMemcpyAsync(stage_.M.p, M, batch * sizeof(Jac), hipMemcpyHostToDevice, stage_.stream));
  CPX_HIP(hipEventRecord(stage_.uploaded, stage_.stream));
  stage_.batch = batch;
  stage_.ell = ell_;
}
void Engine::batch_load_end() {
  if (!stage_.batch) throw std::logic_error("batch_load_begin first");
  if (stage_.ell != ell_) {
    stage_.batch = 0;
    throw std::logic_error("the CRS changed between batch_load_begin and batch_load_end");
  }
  CPX_HIP(hipSetDevice(device_));
  CPX_HIP(hipEventSynchronize(stage_.uploaded));   // the caller's host buffers are free again when this call returns
  const size_t batch = stage_.batch;
  stage_.batch = 0;
  load_rows(batch, reinterpret_cast<const uint8_t*>(stage_.R.p), reinterpret_cast<const uint8_t*>(stage_.S.p), reinterpret_cast<const uint8_t*>(stage_.T.p),
            reinterpret_cast<const uint8_t*>(stage_.U.p), reinterpret_cast<const uint8_t*>(stage_.M.p), true);
  CPX_HIP(hipEventRecord(stage_.consumed, stream_));
  stage_.have_consumed = true;
}

// ---------------------------------------------------------------- prover
namespace {
struct ProverState {
  Transcript tr{"curdleproofs"};
  SVec vec_a, a_perm, x;            // x: SameMSM witness vector
  SVec c, d;                        // IPA vectors
  SVec factors;
  S alpha_sp, beta_sp, gprod, alpha_g, beta_g, beta_g_inv, r_p, z_ip, alpha_i, beta_i, alpha_m;
  SVec vec_c_blinders, rb_plus_alpha, u;
  SVec SG, SGp, SM;                 // all-MSM prover: per-original-base fold coefficients (IPA G, IPA G' incl. u, SameMSM)
  SVec scal[8];                     // scalar vectors handed to the device for the current phase
  S gam[2];
  std::vector<uint8_t> comp;        // compressed bytes of every slot (count * 48)
  S r_p_out, c_final, d_final, z_k, z_t, z_u, x_final;
  const uint8_t* rnd;               // 3n+9 Fr draws, wire form
  S draw(size_t i) const { return S_from_wire(rnd + 32 * i); }
};
}  // namespace

void Engine::batch_prove(const uint32_t* permutation, const uint8_t* k_in, const uint8_t* m_blinders, const uint8_t* rand, uint8_t* proofs_out) {
  if (!B_) throw std::logic_error("batch_load first");
  if (device_prefix(B_)) batch_prove_device(permutation, k_in, m_blinders, rand, proofs_out);   // the whole protocol on the GPU (engine_device.cpp)
  else batch_prove_tables(permutation, k_in, m_blinders, rand, proofs_out);                  // a few proofs: host-driven Fiat-Shamir
}

// ---------------------------------------------------------------- all-MSM prover over shifted-base tables
// Same protocol, same outputs, different evaluation order of the group arithmetic.  The reference folds the
// bases every round (inner_product_argument.rs:174-179, same_multiscalar_argument.rs:126-131) and takes MSMs
// over the folded bases.  A folded base is a known linear combination of ORIGINAL bases,
//     G^(j)_i = sum_t S^(j)_t * G_(t * n/2^j + i),   S^(j)_t = prod_{m<=j} gamma_m^(bit_(j-m) of t)
// (doc/optimizations.md "IPA verification scalars"), so every cross term L/R of round j is an MSM over the n/2
// original bases whose index has bit (L-j) set / clear, with scalars  vector_entry * S  (and * u_k for the
// rescaled basis G' = u o G of grand_product_argument.rs:90-102, which is never materialised).  Likewise every
// commitment (B, D, cm_T, cm_A, A', ...) expands into an MSM over CRS / instance points.  Hence: no basis
// folds, no per-round normalisation, every MSM runs on pre-shifted tables (k_msm_tbl) without a doubling tail.
// The side and table streams of the host-driven prover: their kernels (a few long single-wave groups: the doubling chains of the table
// build and of the MSM tails) run BESIDE the phases of the main stream, and the dispatcher likes to put the main stream's single-wave
// groups onto the very SIMDs those long waves occupy — two 200-VGPR waves then time-share one SIMD and a 120 us reduction took
// 370 us.  The queues of these streams are therefore confined to the upper half of the CU mask (hipExtStreamCreateWithCUMask); the
// main stream, unconfined, fills the GPU from the low end.  Option cu_mask = 0 creates plain streams (A/B runs).
hipStream_t Engine::create_masked_stream(bool upper) {
  hipStream_t st = nullptr;
  hipDeviceProp_t prop;
  CPX_HIP(hipGetDeviceProperties(&prop, device_));
  const int ncu = prop.multiProcessorCount;
  if (!opt_.cu_mask || ncu < 64) {
    CPX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return st;
  }
  std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
  for (int cu = upper ? ncu / 2 : 0; cu < (upper ? ncu : ncu / 2); cu++) mask[(size_t)cu / 32] |= 1u << (cu % 32);
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
    (void)hipGetLastError();
    CPX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  }
  return st;
}
namespace {
// the engine's main stream replaced for the lifetime of the guard; work already queued on the old stream is waited for, and the old
// stream waits for the new one at the end
struct StreamSwap {
  hipStream_t& ref;
  hipStream_t saved;
  hipEvent_t ev;
  StreamSwap(hipStream_t& r, hipStream_t s, hipEvent_t e) : ref(r), saved(r), ev(e) {
    if (!s) return;
    (void)hipEventRecord(ev, saved);
    (void)hipStreamWaitEvent(s, ev, 0);
    ref = s;
  }
  ~StreamSwap() {
    if (ref == saved) return;
    (void)hipEventRecord(ev, ref);
    (void)hipStreamWaitEvent(saved, ev, 0);
    ref = saved;
  }
};
}  // namespace

void Engine::batch_prove_tables(const uint32_t* permutation, const uint8_t* k_in, const uint8_t* m_blinders, const uint8_t* rand, uint8_t* proofs_out) {
  HostSpan wall(this, "host_prove_wall");
  CPX_HIP(hipSetDevice(device_));
  const size_t B = B_, ell = ell_, n = n_, L = L_;
  if (!side_.lat_stream) {
    side_.lat_stream = create_masked_stream(true);
    side_.lat_main = create_masked_stream(false);
    CPX_HIP(hipEventCreateWithFlags(&side_.lat_ev, hipEventDisableTiming));
  }
  const hipStream_t sside = side_.lat_stream;
  TeamScope team(this, B);   // 2 ... device_min_batch - 1 (55) proofs: the host loops between the phases on spinning helper threads
  StreamSwap lat_main(stream_, B <= 8 ? side_.lat_main : nullptr, side_.lat_ev);   // a few proofs: the phases on the lower half of the CUs
  const SlotMap sm(L);
  const size_t nrand = 3 * n + 9;
  const int NS = sm.count();
  const size_t NP = np();
  std::vector<ProverState> st(B);
  std::vector<uint8_t> comp;
  const size_t I_AB = 0, I_CB = 2, I_IR = 6, I_IZ = 6 + n, I_RT = 2 * n + 4, I_RU = 2 * n + 5, I_RA = 2 * n + 6, I_RB = 2 * n + 7,
               I_RK = 2 * n + 8, I_VR = 2 * n + 9;
  // table offsets inside a per-proof row
  const size_t O_M = 0, O_T = 1, O_U = 1 + n;
  // CRS table indices: G | Hvec at 0..n-1, H at n, G_t at n+1, G_u at n+2
  const uint32_t* iH = idx_list({(uint32_t)n});
  const uint32_t* iGt = idx_list({(uint32_t)(n + 1)});
  const uint32_t* iGu = idx_list({(uint32_t)(n + 2)});
  std::vector<uint32_t> gb(n);   // SameMSM basis G | Hvec[0..2) | G_t | G_u as CRS-table indices
  for (size_t i = 0; i < ell + 2; i++) gb[i] = (uint32_t)i;
  gb[ell + 2] = (uint32_t)(n + 1);
  gb[ell + 3] = (uint32_t)(n + 2);
  const uint32_t* iGb = idx_list(gb);
  const TblSeg none{nullptr, nullptr, 0, 0};

  // -- P0: compressed instance vectors, M -> affine (into table slot 0), then the per-proof tables.  The transcript
  //    prefix (instance + M absorbed, vec_a drawn) is hashed on the side stream while the tables are built.
  h_inst_comp_.ensure(B * 4 * ell * 48);   // pinned and persistent: no page faults, true async copies
  h_mcomp_.ensure(B * 48);
  h_u32_.ensure(B);
  const uint8_t* inst_comp = h_inst_comp_.p;
  {
    d_bytes_.ensure(B * 4 * ell * 48);
    tick("k_compress", 0, (double)(4 * ell * B));
    launch_compress(d_pp_.p, (int)(4 * ell), (int)pp_stride_, (int)B, d_bytes_.p, stream_);
    tock();
    d_dst_.ensure(B);
    d_mcomp_.ensure(B * 48);
    uint32_t* dst = h_u32_.p;
    for (size_t p = 0; p < B; p++) dst[p] = (uint32_t)(p * NP + O_M);
    CPX_HIP(hipMemcpyAsync(d_dst_.p, dst, B * 4, hipMemcpyHostToDevice, stream_));
    launch_finalize(d_Mjac_.p, (int)B, d_psrc_.p, d_dst_.p, d_mcomp_.p, stream_);
    transcript_prefix_async(B);   // side stream: copies of the compressed bytes for the host's transcripts
    // the per-proof tables on the table stream: M's row first (phase 2 needs it: B = A + alpha M + ...), then T and U (needed from
    // SameMSM step 1 on), beside phase 1 instead of in front of it
    if (!tab_.stream) {
      tab_.stream = create_masked_stream(true);
      CPX_HIP(hipEventCreateWithFlags(&tab_.ev_start, hipEventDisableTiming));
      CPX_HIP(hipEventCreateWithFlags(&tab_.ev_m, hipEventDisableTiming));
      CPX_HIP(hipEventCreateWithFlags(&tab_.ev_done, hipEventDisableTiming));
    }
    d_tbltmp_.ensure(B * NP * (size_t)(copies_ / 2 - 1));   // (all rows in two launches; load_batch sized the scratch for the device prover's chunks: a no-op below 3072 proofs)
    CPX_HIP(hipEventRecord(tab_.ev_start, stream_));
    CPX_HIP(hipStreamWaitEvent(tab_.stream, tab_.ev_start, 0));
    const size_t tmp_m = B * (size_t)(copies_ / 2 - 1);   // scratch entries of the M launch; the T | U launch takes the rest
    launch_table_build(opt_, d_psrc_.p, NP, d_ptab_.p, (int)B, (size_t)copies_ * NP, 1, (int)NP, copies_, true, d_tbltmp_.p, tab_.stream);
    CPX_HIP(hipEventRecord(tab_.ev_m, tab_.stream));
    launch_table_build(opt_, d_psrc_.p + 1, NP, d_ptab_.p + 1, (int)B, (size_t)copies_ * NP, (int)(2 * n), (int)NP, copies_, true, d_tbltmp_.p + tmp_m, tab_.stream);
    const uint8_t* mcomp = h_mcomp_.p;
    wait_side();
    parallel_for(B, [&](size_t p) {
      ProverState& s = st[p];
      s.rnd = rand + p * nrand * 32;
      s.comp.assign((size_t)NS * 48, 0);
      memcpy(&s.comp[SL_M * 48], &mcomp[p * 48], 48);
      {   // curdleproofs.rs:78-83
        const uint8_t* ic = &inst_comp[p * 4 * ell * 48];
        for (int v = 0; v < 4; v++) s.tr.append_point_vec_bytes("curdleproofs_step1", ic + v * ell * 48, ell);
        s.tr.append_point_bytes("curdleproofs_step1", &s.comp[SL_M * 48]);
        s.vec_a = s.tr.get_and_append_challenges("curdleproofs_vec_a", ell);
      }
      const uint32_t* perm = permutation + p * ell;
      s.a_perm.resize(ell);
      for (size_t i = 0; i < ell; i++) s.a_perm[i] = s.vec_a[perm[i]];
      const S kk = S_from_wire(k_in + 32 * p), r_k = s.draw(I_RK);
      // scalar vectors of phase 1
      s.scal[0] = s.a_perm;                                   // A (and the head of A') over G | Hvec
      s.scal[0].push_back(s.draw(I_AB));
      s.scal[0].push_back(s.draw(I_AB + 1));
      s.scal[0].push_back(S::zero());
      s.scal[0].push_back(S::zero());
      s.scal[0].push_back(s.draw(I_RT));                      // ... then r_t on G_t, r_u on G_u for A' (curdleproofs.rs:134)
      s.scal[0].push_back(s.draw(I_RU));
      s.scal[1].resize(n);                                    // vec_r of SameMSM: B_a, B_t, B_u
      for (size_t i = 0; i < n; i++) s.scal[1][i] = s.draw(I_VR + i);
      s.scal[2] = {s.draw(I_RT), s.draw(I_RU), s.draw(I_RA), s.draw(I_RB)};
      s.scal[3] = {kk, r_k};                                   // scalars of the four T_2 scalar multiplications
      s.scal[4].resize(n);                                     // r_c of the IPA (inner_product_argument.rs:46): B_c = msm(G | Hvec, r_c)
      for (size_t i = 0; i < n; i++) s.scal[4][i] = s.draw(I_IR + i);
    });
  }
  auto take = [&](size_t p, int slot_id, size_t req_index) { memcpy(&st[p].comp[(size_t)slot_id * 48], &comp[req_index * 48], 48); };

  // -- side stream: R = a x vec_R and S = a x vec_S (curdleproofs.rs:112-113).  The instance points R_i, S_i are
  //    used by exactly these MSMs, so they get no table: the endomorphism bucket-list kernel (the verifier's) + tails off the
  //    critical path; the affine results land in the slots SL_R / SL_S.
  //    The four scalar multiplications k R, k S (curdleproofs.rs:115-116), r_k R, r_k S (same_scalar_argument.rs:60-61) ride along
  //    as four more MSMs over the same points with the scalars k a and r_k a: k (a x vec_R) = (k a) x vec_R.  A 255-bit
  //    double-and-add chain of one lane (k_smul: 4.6 ms) was what a lone proof waited for at the SameScalar step; as MSM tasks
  //    they run beside R and S and finish with them.
  const int CW0 = SL_CMT1, CWN = sm.CMB2() - SL_CMT1 + 1;   // slot window compressed on the side stream
  {
    const size_t nt = 6 * B, total = B * ell;
    side_.scal.ensure(3 * total);
    side_.tasks.ensure(nt);
    side_.res.ensure(nt);
    side_.dst.ensure(2 * nt);
    side_.comp.ensure(B * (size_t)CWN * 48);
    side_.hcomp.ensure(B * (size_t)CWN * 48);
    const size_t o_tasks = 3 * total * sizeof(Fr), o_dst = o_tasks + nt * sizeof(MsmTask);
    side_.stage.ensure(o_dst + 2 * nt * sizeof(uint32_t));
    Fr* hs = reinterpret_cast<Fr*>(side_.stage.p);
    MsmTask* ht = reinterpret_cast<MsmTask*>(side_.stage.p + o_tasks);
    uint32_t* hd = reinterpret_cast<uint32_t*>(side_.stage.p + o_dst);   // [nt] destination slots, [nt] addend slots
    parallel_for(B, [&](size_t p) {
      const S kk = st[p].scal[3][0], r_k = st[p].scal[3][1];
      for (size_t i = 0; i < ell; i++) {
        hs[p * ell + i] = st[p].vec_a[i].f;
        hs[total + p * ell + i] = (st[p].vec_a[i] * kk).f;
        hs[2 * total + p * ell + i] = (st[p].vec_a[i] * r_k).f;
      }
      // tasks [0, 2B): R, S; tasks [2B, 6B): k R, k S, r_k R, r_k S (finalised later, on top of the r H points of phase 1)
      const int dsts[6] = {SL_R, SL_S, SL_CMT2, SL_CMU2, sm.CMA2(), sm.CMB2()};
      for (int j = 0; j < 6; j++) {
        const size_t t = j < 2 ? 2 * p + j : 2 * B + 4 * p + (j - 2);
        const Fr* sc = side_.scal.p + (size_t)(j / 2) * total + p * ell;
        ht[t] = MsmTask{pp(p) + (j & 1) * ell, nullptr, sc, (uint32_t)ell, 0, (uint32_t)(t * ell)};
        hd[t] = slot_index(p, dsts[j]);
        hd[nt + t] = j < 2 ? ~0u : slot_index(p, sm.TMP(j - 2));
      }
    });
    CPX_HIP(hipMemcpyAsync(side_.scal.p, hs, 3 * total * sizeof(Fr), hipMemcpyHostToDevice, sside));
    CPX_HIP(hipMemcpyAsync(side_.tasks.p, ht, nt * sizeof(MsmTask), hipMemcpyHostToDevice, sside));
    CPX_HIP(hipMemcpyAsync(side_.dst.p, hd, 2 * nt * sizeof(uint32_t), hipMemcpyHostToDevice, sside));
    const size_t slices = (size_t)msm_tblw_slices(opt_, (int)nt, 2, (int)ell), nsets = nt * 32 * slices;
    side_.conv.ensure(2 * nt * ell);   // per task: points and images
    side_.ttasks.ensure(nt);
    side_.digits.ensure(9 * nt * ell);
    side_.raw.ensure(nsets * raw_set_words());
    side_.rawslot.ensure(nsets);
    side_.mid.ensure(nsets * reduce_mid_per_set());
    side_.part.ensure(nsets);
    launch_msm_endo(side_.tasks.p, (int)nt, (int)ell, side_.conv.p, side_.digits.p, side_.ttasks.p, side_.raw.p, side_.rawslot.p, sside, (int)slices);
    launch_reduce_sets(opt_, side_.raw.p, side_.rawslot.p, 0, (int)nsets, side_.mid.p, side_.part.p, sside, (int)B);
    launch_msm_tail(opt_, side_.part.p, nullptr, side_.res.p, (int)nt, 16, 8, sside, nullptr, 0, (int)(2 * slices));
    launch_finalize(side_.res.p, (int)(2 * B), d_pp_.p, side_.dst.p, nullptr, sside);
  }

  // -- table stream: B_t = msm(T_b, vec_r), B_u = msm(U_b, vec_r) (same_multiscalar_argument.rs:81-82) right behind the tables of T and U
  {
    const size_t nt = 2 * B;
    const int wpw = msm_tblw_windows_per_wave(opt_, (int)nt), tslices = msm_tblw_slices(opt_, (int)nt, wpw, (int)n);
    const uint32_t parts = (uint32_t)(msm_tblw_parts(wpw) * tslices);
    const size_t b_scal = B * n * sizeof(Fr), b_tt = nt * sizeof(TblTask), b_blob = b_scal + b_tt + 3 * nt * sizeof(uint32_t);
    tab_.blob.ensure(b_blob);
    tab_.stage.ensure(b_blob);
    tab_.raw.ensure(nt * parts * raw_set_words());
    tab_.rawslot.ensure(nt * parts);
    tab_.mid.ensure(nt * parts * reduce_mid_per_set());
    tab_.part.ensure(nt * parts);
    tab_.comp.ensure(nt * 48);
    tab_.hcomp.ensure(nt * 48);
    Fr* hs = reinterpret_cast<Fr*>(tab_.stage.p);
    TblTask* ht = reinterpret_cast<TblTask*>(tab_.stage.p + b_scal);
    uint32_t* hu = reinterpret_cast<uint32_t*>(tab_.stage.p + b_scal + b_tt);
    const Fr* d_scal = reinterpret_cast<const Fr*>(tab_.blob.p);
    const uint32_t dummy_dst = slot_index(0, sm.TMP(7));
    for (size_t p = 0; p < B; p++) {
      for (size_t i = 0; i < n; i++) hs[p * n + i] = st[p].scal[1][i].f;
      for (int q = 0; q < 2; q++) {
        TblTask t;
        t.seg[0] = pseg(p, q ? O_U : O_T, (uint32_t)n);
        t.seg[1] = none;
        t.scalars = d_scal + p * n;
        t.flags = 0;
        t.pad = (uint32_t)((2 * p + q) * parts);
        t.digits = nullptr;
        ht[2 * p + q] = t;
        hu[2 * p + q] = (uint32_t)((2 * p + q) * parts);
        hu[nt + 2 * p + q] = parts;
        hu[2 * nt + 2 * p + q] = dummy_dst;
      }
    }
    CPX_HIP(hipMemcpyAsync(tab_.blob.p, tab_.stage.p, b_blob, hipMemcpyHostToDevice, tab_.stream));
    const uint32_t* d_u32 = reinterpret_cast<const uint32_t*>(tab_.blob.p + b_scal + b_tt);
    launch_msm_tblw(reinterpret_cast<const TblTask*>(tab_.blob.p + b_scal), (int)nt, wpw, tab_.raw.p, tab_.rawslot.p, tab_.stream, tslices);
    launch_reduce_sets(opt_, tab_.raw.p, tab_.rawslot.p, 0, (int)(nt * parts), tab_.mid.p, tab_.part.p, tab_.stream, (int)B);
    launch_finalize_ranges(opt_, tab_.part.p, d_u32, d_u32 + nt, (int)nt, d_pp_.p, d_u32 + 2 * nt, tab_.comp.p, tab_.stream);
    CPX_HIP(hipMemcpyAsync(tab_.hcomp.p, tab_.comp.p, nt * 48, hipMemcpyDeviceToHost, tab_.stream));
    CPX_HIP(hipEventRecord(tab_.ev_done, tab_.stream));
  }

  // -- P1: everything that depends only on vec_a and the prover's randomness
  {
    std::vector<TblReq> reqs;
    for (size_t p = 0; p < B; p++) {
      ProverState& s = st[p];
      const S* rt = &s.scal[2][0];
      const S* ru = &s.scal[2][1];
      const S* ra = &s.scal[2][2];
      const S* rb = &s.scal[2][3];
      reqs.push_back({cseg(0, (uint32_t)n), s.scal[0].data(), none, nullptr, slot_index(p, SL_A)});  // A (kept as a point: B and A' add it)
      reqs.push_back({cseg(0, (uint32_t)n, iGb), s.scal[1].data(), none, nullptr});                   // B_a (same_multiscalar_argument.rs:80)
      reqs.push_back({cseg(0, 1, iGt), rt, none, nullptr, slot_index(p, SL_CMT1)});                   // cm_T.T_1 = r_t G_t
      reqs.push_back({cseg(0, 1, iGu), ru, none, nullptr, slot_index(p, SL_CMU1)});                   // cm_U.T_1
      reqs.push_back({cseg(0, 1, iGt), ra, none, nullptr});                                           // cm_A.T_1
      reqs.push_back({cseg(0, 1, iGu), rb, none, nullptr});                                           // cm_B.T_1
      // r * H halves of the four T_2 commitments, kept as affine points for the side-stream scalar multiplications
      reqs.push_back({cseg(0, 1, iH), rt, none, nullptr, slot_index(p, sm.TMP(0))});
      reqs.push_back({cseg(0, 1, iH), ru, none, nullptr, slot_index(p, sm.TMP(1))});
      reqs.push_back({cseg(0, 1, iH), ra, none, nullptr, slot_index(p, sm.TMP(2))});
      reqs.push_back({cseg(0, 1, iH), rb, none, nullptr, slot_index(p, sm.TMP(3))});
      reqs.push_back({cseg(0, (uint32_t)n), s.scal[4].data(), none, nullptr});                        // B_c: depends on the randomness only
    }
    run_tbl_phase(reqs, &comp);
    parallel_for(B, [&](size_t p) {
      ProverState& s = st[p];
      const int ids[11] = {SL_A, sm.BA(), SL_CMT1, SL_CMU1, sm.CMA1(), sm.CMB1(), -1, -1, -1, -1, SL_BC};
      for (int i = 0; i < 11; i++)
        if (ids[i] >= 0) take(p, ids[i], p * 11 + i);
      // same_permutation_argument.rs:60-83
      s.tr.append_point_bytes("same_perm_step1", &s.comp[SL_A * 48]);
      s.tr.append_point_bytes("same_perm_step1", &s.comp[SL_M * 48]);
      s.tr.append_scalar_vec("same_perm_step1", s.vec_a);
      s.alpha_sp = s.tr.get_and_append_challenge("same_perm_alpha");
      s.beta_sp = s.tr.get_and_append_challenge("same_perm_beta");
      const uint32_t* perm = permutation + p * ell;
      s.factors.resize(ell);
      s.gprod = S::one();
      for (size_t i = 0; i < ell; i++) {
        s.factors[i] = s.a_perm[i] + S::from_u64(perm[i]) * s.alpha_sp + s.beta_sp;
        s.gprod *= s.factors[i];
      }
      // B = A + alpha M + beta * sum(G) (same_permutation_argument.rs:75-76): the point A of phase 1 plus two
      // single-point terms (G_sum sits in the CRS tables)
      s.scal[5] = {s.beta_sp};
      s.scal[6] = {s.alpha_sp};
      // the partial products c (grand_product_argument.rs:66-75) need the factors only, so C joins the phase of B and A'
      s.c.assign(1, S::one());
      for (size_t i = 0; i + 1 < ell; i++) s.c.push_back(s.c[i] * s.factors[i]);
      s.vec_c_blinders.resize(N_BLINDERS);
      for (size_t i = 0; i < N_BLINDERS; i++) s.vec_c_blinders[i] = s.draw(I_CB + i);
      s.c.insert(s.c.end(), s.vec_c_blinders.begin(), s.vec_c_blinders.end());
    });
  }

  // -- side stream: cm_T.T_2 = k R + r_t H, cm_U.T_2 = k S + r_u H (curdleproofs.rs:115-116), cm_A.T_2 = r_k R + r_a H,
  //    cm_B.T_2 = r_k S + r_b H (same_scalar_argument.rs:60-61): the four MSM results of the side stream on top of the r*H points
  //    phase 1 left in TMP0..3 (complete by now: the host has taken phase 1's results); needed only at the SameScalar transcript step.
  {
    const size_t nt = 6 * B;
    launch_finalize(side_.res.p + 2 * B, (int)(4 * B), d_pp_.p, side_.dst.p + 2 * B, nullptr, sside, side_.dst.p + nt + 2 * B);
    launch_compress(d_pp_.p + 4 * ell + CW0, CWN, (int)pp_stride_, (int)B, side_.comp.p, sside);
    CPX_HIP(hipMemcpyAsync(side_.hcomp.p, side_.comp.p, B * (size_t)CWN * 48, hipMemcpyDeviceToHost, sside));
    CPX_HIP(hipEventRecord(side_.ev, sside));
  }

  // -- P2: B, A' = A + cm_T.T_1 + cm_U.T_1 (curdleproofs.rs:134) as a sum of three points of phase 1, and C = msm(G | Hvec, c)
  //    (grand_product_argument.rs:76).  The transcript takes B, draws alpha, then takes C: an order of hashing, not of computing.
  {
    std::vector<TblReq> reqs;
    for (size_t p = 0; p < B; p++) {
      TblReq rb{cseg(n + 3, 1), st[p].scal[5].data(), pseg(p, O_M, 1), st[p].scal[6].data(), slot_index(p, SL_B)};
      rb.add[0] = slot_index(p, SL_A);
      TblReq ra{none, nullptr, none, nullptr};
      ra.add[0] = slot_index(p, SL_A);
      ra.add[1] = slot_index(p, SL_CMT1);
      ra.add[2] = slot_index(p, SL_CMU1);
      reqs.push_back(rb);
      reqs.push_back(ra);
      reqs.push_back({cseg(0, (uint32_t)n), st[p].c.data(), none, nullptr});
    }
    CPX_HIP(hipStreamWaitEvent(stream_, tab_.ev_m, 0));   // M's table row (table stream)
    run_tbl_phase(reqs, &comp);
    parallel_for(B, [&](size_t p) {
      ProverState& s = st[p];
      take(p, SL_B, p * 3);
      take(p, sm.APRIME(), p * 3 + 1);
      take(p, SL_C, p * 3 + 2);
      s.tr.append_point_bytes("gprod_step1", &s.comp[SL_B * 48]);
      s.tr.append_scalar("gprod_step1", s.gprod);
      s.alpha_g = s.tr.get_and_append_challenge("gprod_alpha");
      const S mb[4] = {S_from_wire(m_blinders + (p * 4 + 0) * 32), S_from_wire(m_blinders + (p * 4 + 1) * 32),
                       S_from_wire(m_blinders + (p * 4 + 2) * 32), S_from_wire(m_blinders + (p * 4 + 3) * 32)};
      const S ab[4] = {s.draw(I_AB), s.draw(I_AB + 1), S::zero(), S::zero()};
      s.rb_plus_alpha.resize(N_BLINDERS);
      for (size_t i = 0; i < N_BLINDERS; i++) s.rb_plus_alpha[i] = (ab[i] + s.alpha_sp * mb[i]) + s.alpha_g;
      s.r_p = host::inner_product(s.rb_plus_alpha.data(), s.vec_c_blinders.data(), N_BLINDERS);
      s.tr.append_point_bytes("gprod_step2", &s.comp[SL_C * 48]);
      s.tr.append_scalar("gprod_step2", s.r_p);
      s.beta_g = s.tr.get_and_append_challenge("gprod_beta");
      s.beta_g_inv = s.beta_g.inverse();
      s.u.resize(n);
      S pw = s.beta_g_inv;
      for (size_t i = 0; i < ell; i++) {
        s.u[i] = pw;
        pw *= s.beta_g_inv;
      }
      for (size_t i = ell; i < n; i++) s.u[i] = pw;
      s.d.resize(n);
      S pb = s.beta_g, pbm = S::one();
      for (size_t i = 0; i < ell; i++) {
        s.d[i] = s.factors[i] * pb - pbm;
        pbm = pb;
        pb *= s.beta_g;
      }
      const S beta_l1 = pbm * s.beta_g, beta_l = pbm;
      for (size_t i = 0; i < N_BLINDERS; i++) s.d[ell + i] = beta_l1 * s.rb_plus_alpha[i];
      s.z_ip = s.r_p * beta_l1 + s.gprod * beta_l - S::one();
      // D = B - beta^-1 sum(G) + alpha sum(Hvec) (grand_product_argument.rs:132): the point B plus two single-point terms
      s.scal[0] = {-s.beta_g_inv, s.alpha_g};
      // generate_ipa_blinders (inner_product_argument.rs:42-82)
      SVec r(n), zz(n);
      for (size_t i = 0; i < n; i++) r[i] = s.draw(I_IR + i);
      for (size_t i = 0; i + 2 < n; i++) zz[i] = s.draw(I_IZ + i);
      const S omega = host::inner_product(r.data(), s.d.data(), n) + host::inner_product(zz.data(), s.c.data(), n - 2);
      const S delta = host::inner_product(r.data(), zz.data(), n - 2);
      const S inv_c = s.c[n - 2].inverse();
      const S last_z = (r[n - 2] * inv_c * omega - delta) * ((-r[n - 2]) * inv_c * s.c[n - 1] + r[n - 1]).inverse();
      const S pen_z = (-inv_c) * (last_z * s.c[n - 1] + omega);
      zz[n - 2] = pen_z;
      zz[n - 1] = last_z;
      s.scal[1] = r;                                      // r_c -> B_c over G
      s.scal[2] = zz;                                     // r_d
      s.scal[3].resize(n);                                // B_d = msm(G', r_d) = msm(G, r_d o u)
      for (size_t i = 0; i < n; i++) s.scal[3][i] = zz[i] * s.u[i];
    });
  }

  // -- P3: D, B_d
  h_rvec_.ensure(B * 4 * n);
  h_rfin_.ensure(B * 3);   // c_final, d_final, x_final of every proof
  h_rgam_.ensure(B * 2);
  d_rvec_.ensure(B * 4 * n);
  d_rgam_.ensure(B * 2);
  d_rbeta_.ensure(B);
  d_rout_.ensure(B * (2 * n + 2));
  {
    std::vector<TblReq> reqs;
    for (size_t p = 0; p < B; p++) {
      ProverState& s = st[p];
      TblReq rd{cseg(n + 3, 2), s.scal[0].data(), none, nullptr};   // G_sum, H_sum are columns n+3, n+4 of the CRS tables
      rd.add[0] = slot_index(p, SL_B);
      reqs.push_back(rd);
      reqs.push_back({cseg(0, (uint32_t)n), s.scal[3].data(), none, nullptr});
    }
    run_tbl_phase(reqs, &comp);
    parallel_for(B, [&](size_t p) {
      ProverState& s = st[p];
      take(p, sm.D(), p * 2);
      take(p, SL_BD, p * 2 + 1);
      s.tr.append_point_bytes("ipa_step1", &s.comp[SL_C * 48]);
      s.tr.append_point_bytes("ipa_step1", &s.comp[sm.D() * 48]);
      s.tr.append_scalar("ipa_step1", s.z_ip);
      s.tr.append_point_bytes("ipa_step1", &s.comp[SL_BC * 48]);
      s.tr.append_point_bytes("ipa_step1", &s.comp[SL_BD * 48]);
      s.alpha_i = s.tr.get_and_append_challenge("ipa_alpha");
      s.beta_i = s.tr.get_and_append_challenge("ipa_beta");
      for (size_t i = 0; i < n; i++) {
        s.c[i] = s.scal[1][i] + s.alpha_i * s.c[i];
        s.d[i] = s.scal[2][i] + s.alpha_i * s.d[i];
      }
      s.SG.assign(n, S::one());
      s.SGp = s.u;
      // the round vectors live on the device from here on: c | d | SG | SGp
      Fr* v = h_rvec_.p + p * 4 * n;
      for (size_t i = 0; i < n; i++) {
        v[i] = s.c[i].f;
        v[n + i] = s.d[i].f;
        v[2 * n + i] = s.SG[i].f;
        v[3 * n + i] = s.SGp[i].f;
      }
      h_rgam_.p[p] = s.beta_i.f;
    });
    CPX_HIP(hipMemcpyAsync(d_rvec_.p, h_rvec_.p, B * 4 * n * sizeof(Fr), hipMemcpyHostToDevice, stream_));
    CPX_HIP(hipMemcpyAsync(d_rbeta_.p, h_rgam_.p, B * sizeof(Fr), hipMemcpyHostToDevice, stream_));
    wait_stream();   // h_rgam_ is reused for the round challenges
  }

  // -- P5: IPA rounds as MSMs over the original bases.  The Fr side of a round (cross-term scalars, inner products,
  //    folds of c and d, fold-coefficient updates) runs on the device (k_ipa_round_scalars / k_ipa_round_fold); the
  //    host hashes the four points of the round and returns gamma, gamma^-1.
  for (size_t j = 0; j < L; j++) {
    const size_t half = n >> (j + 1);
    std::vector<uint32_t> hi, lo;
    for (size_t k = 0; k < n; k++) ((k & half) ? hi : lo).push_back((uint32_t)k);
    const uint32_t* d_hi = idx_list(hi);
    const uint32_t* d_lo = idx_list(lo);
    const size_t hn = n / 2, per = 4 * hn + 2;
    launch_ipa_round_scalars(d_rvec_.p, (int)B, (int)n, (int)half, d_rbeta_.p, d_rout_.p, stream_);
    std::vector<TblReq> reqs;
    for (size_t p = 0; p < B; p++) {
      const Fr* o = d_rout_.p + p * per;   // [L_C (hn), beta<c_L,d_R> | L_D (hn) | R_C (hn), beta<c_R,d_L> | R_D (hn)]
      TblReq r0{cseg(0, (uint32_t)hn, d_hi), nullptr, cseg(0, 1, iH), nullptr};
      r0.dev = o;
      TblReq r1{cseg(0, (uint32_t)hn, d_lo), nullptr, none, nullptr};
      r1.dev = o + hn + 1;
      TblReq r2{cseg(0, (uint32_t)hn, d_lo), nullptr, cseg(0, 1, iH), nullptr};
      r2.dev = o + 2 * hn + 1;
      TblReq r3{cseg(0, (uint32_t)hn, d_hi), nullptr, none, nullptr};
      r3.dev = o + 3 * hn + 2;
      reqs.push_back(r0);
      reqs.push_back(r1);
      reqs.push_back(r2);
      reqs.push_back(r3);
    }
    run_tbl_phase(reqs, &comp);
    parallel_for(B, [&](size_t p) {
      ProverState& s = st[p];
      const int order[4] = {sm.LC((int)j), sm.LD((int)j), sm.RC((int)j), sm.RD((int)j)};
      for (int q = 0; q < 4; q++) take(p, order[q], p * 4 + q);
      for (int q : order) s.tr.append_point_bytes("ipa_loop", &s.comp[(size_t)q * 48]);
      const S gamma = s.tr.get_and_append_challenge("ipa_gamma");
      h_rgam_.p[2 * p] = gamma.f;
      h_rgam_.p[2 * p + 1] = gamma.inverse().f;
    });
    CPX_HIP(hipMemcpyAsync(d_rgam_.p, h_rgam_.p, B * 2 * sizeof(Fr), hipMemcpyHostToDevice, stream_));
    launch_ipa_round_fold(d_rvec_.p, (int)B, (int)n, (int)half, d_rgam_.p, stream_);
    if (j + 1 == L) {   // c_final, d_final (inner_product_argument.rs:188-195)
      CPX_HIP(hipMemcpy2DAsync(h_rfin_.p, 3 * sizeof(Fr), d_rvec_.p, 4 * n * sizeof(Fr), sizeof(Fr), B, hipMemcpyDeviceToHost, stream_));
      CPX_HIP(hipMemcpy2DAsync(h_rfin_.p + 1, 3 * sizeof(Fr), d_rvec_.p + n, 4 * n * sizeof(Fr), sizeof(Fr), B, hipMemcpyDeviceToHost, stream_));
      wait_stream();
    }
    // no sync otherwise: h_rgam_ is rewritten only after the next round's run_tbl_phase has synchronised the stream
  }

  // -- P6 (host only): SameScalar transcript, SameMSM step 1
  {
    std::vector<uint8_t> id_comp(48, 0);
    id_comp[0] = 0xc0;
    wait_side();   // R, S and the four T_2 commitments from the side stream
    {
      HostSpan w(this, "host_wait_device");
      CPX_HIP(hipEventSynchronize(tab_.ev_done));   // the tables of T and U, B_t and B_u from the table stream
    }
    CPX_HIP(hipStreamWaitEvent(stream_, tab_.ev_done, 0));
    parallel_for(B, [&](size_t p) {
      ProverState& s = st[p];
      memcpy(&s.comp[(size_t)sm.BT() * 48], tab_.hcomp.p + (2 * p) * 48, 48);
      memcpy(&s.comp[(size_t)sm.BU() * 48], tab_.hcomp.p + (2 * p + 1) * 48, 48);
      memcpy(s.c_final.f.v, &h_rfin_.p[3 * p], sizeof(Fr));       // c[0], d[0] after the last fold (device-resident vectors)
      memcpy(s.d_final.f.v, &h_rfin_.p[3 * p + 1], sizeof(Fr));
      const int side_slots[6] = {SL_R, SL_S, SL_CMT2, SL_CMU2, sm.CMA2(), sm.CMB2()};
      for (int q : side_slots) memcpy(&s.comp[(size_t)q * 48], side_.hcomp.p + (p * (size_t)CWN + (q - CW0)) * 48, 48);
      const int pts[10] = {SL_R, SL_S, SL_CMT1, SL_CMT2, SL_CMU1, SL_CMU2, sm.CMA1(), sm.CMA2(), sm.CMB1(), sm.CMB2()};
      for (int q : pts) s.tr.append_point_bytes("sameexp_points", &s.comp[(size_t)q * 48]);
      const S alpha = s.tr.get_and_append_challenge("same_scalar_alpha");
      const S kk = S_from_wire(k_in + 32 * p);
      s.z_k = s.draw(I_RK) + kk * alpha;
      s.z_t = s.draw(I_RA) + s.draw(I_RT) * alpha;
      s.z_u = s.draw(I_RB) + s.draw(I_RU) * alpha;
      s.tr.append_point_bytes("same_msm_step1", &s.comp[(size_t)sm.APRIME() * 48]);
      s.tr.append_point_bytes("same_msm_step1", &s.comp[SL_CMT2 * 48]);
      s.tr.append_point_bytes("same_msm_step1", &s.comp[SL_CMU2 * 48]);
      std::vector<uint8_t> vb(n * 48);
      const uint8_t* ic = &inst_comp[p * 4 * ell * 48];
      memcpy(vb.data(), ic + 2 * ell * 48, ell * 48);
      const uint8_t* tailT[4] = {id_comp.data(), id_comp.data(), crs_H_comp_, id_comp.data()};
      for (int i = 0; i < 4; i++) memcpy(&vb[(ell + i) * 48], tailT[i], 48);
      s.tr.append_point_vec_bytes("same_msm_step1", vb.data(), n);
      memcpy(vb.data(), ic + 3 * ell * 48, ell * 48);
      const uint8_t* tailU[4] = {id_comp.data(), id_comp.data(), id_comp.data(), crs_H_comp_};
      for (int i = 0; i < 4; i++) memcpy(&vb[(ell + i) * 48], tailU[i], 48);
      s.tr.append_point_vec_bytes("same_msm_step1", vb.data(), n);
      s.tr.append_point_bytes("same_msm_step1", &s.comp[(size_t)sm.BA() * 48]);
      s.tr.append_point_bytes("same_msm_step1", &s.comp[(size_t)sm.BT() * 48]);
      s.tr.append_point_bytes("same_msm_step1", &s.comp[(size_t)sm.BU() * 48]);
      s.alpha_m = s.tr.get_and_append_challenge("same_msm_alpha");
      s.x.resize(n);
      for (size_t i = 0; i < ell; i++) s.x[i] = s.draw(I_VR + i) + s.alpha_m * s.a_perm[i];
      s.x[ell] = s.draw(I_VR + ell) + s.alpha_m * s.draw(I_AB);
      s.x[ell + 1] = s.draw(I_VR + ell + 1) + s.alpha_m * s.draw(I_AB + 1);
      s.x[ell + 2] = s.draw(I_VR + ell + 2) + s.alpha_m * s.draw(I_RT);
      s.x[ell + 3] = s.draw(I_VR + ell + 3) + s.alpha_m * s.draw(I_RU);
      s.SM.assign(n, S::one());
      Fr* v = h_rvec_.p + p * 2 * n;   // device-resident round vectors: x | SM
      for (size_t i = 0; i < n; i++) {
        v[i] = s.x[i].f;
        v[n + i] = s.SM[i].f;
      }
    });
    CPX_HIP(hipMemcpyAsync(d_rvec_.p, h_rvec_.p, B * 2 * n * sizeof(Fr), hipMemcpyHostToDevice, stream_));
  }

  // -- P7: SameMSM rounds; the Fr side (cross-term scalars, fold of x, fold-coefficient update) on the device
  for (size_t j = 0; j < L; j++) {
    const size_t half = n >> (j + 1);
    std::vector<uint32_t> hi, lo, ghi, glo;
    for (size_t k = 0; k < n; k++) {
      ((k & half) ? hi : lo).push_back((uint32_t)k);
      ((k & half) ? ghi : glo).push_back(gb[k]);
    }
    const uint32_t* d_hi = idx_list(hi);
    const uint32_t* d_lo = idx_list(lo);
    const uint32_t* d_ghi = idx_list(ghi);
    const uint32_t* d_glo = idx_list(glo);
    const size_t hn = n / 2;
    launch_smsm_round_scalars(d_rvec_.p, (int)B, (int)n, (int)half, d_rout_.p, stream_);
    std::vector<TblReq> reqs;
    for (size_t p = 0; p < B; p++) {
      const Fr* oL = d_rout_.p + p * 2 * hn;   // [L_* scalars (hn) | R_* scalars (hn)]
      const Fr* oR = oL + hn;
      TblReq r[6] = {{cseg(0, (uint32_t)hn, d_ghi), nullptr, none, nullptr},       // L_A
                     {pseg(p, O_T, (uint32_t)hn, d_hi), nullptr, none, nullptr},   // L_T
                     {pseg(p, O_U, (uint32_t)hn, d_hi), nullptr, none, nullptr},   // L_U
                     {cseg(0, (uint32_t)hn, d_glo), nullptr, none, nullptr},       // R_A
                     {pseg(p, O_T, (uint32_t)hn, d_lo), nullptr, none, nullptr},   // R_T
                     {pseg(p, O_U, (uint32_t)hn, d_lo), nullptr, none, nullptr}};  // R_U
      for (int q = 0; q < 6; q++) {
        r[q].dev = q < 3 ? oL : oR;
        reqs.push_back(r[q]);
      }
    }
    run_tbl_phase(reqs, &comp);
    parallel_for(B, [&](size_t p) {
      ProverState& s = st[p];
      const int order[6] = {sm.LA((int)j), sm.LT((int)j), sm.LU((int)j), sm.RA((int)j), sm.RT((int)j), sm.RU((int)j)};
      for (int i = 0; i < 6; i++) take(p, order[i], p * 6 + i);
      for (int q : order) s.tr.append_point_bytes("same_msm_loop", &s.comp[(size_t)q * 48]);
      const S gamma = s.tr.get_and_append_challenge("same_msm_gamma");
      h_rgam_.p[2 * p] = gamma.f;
      h_rgam_.p[2 * p + 1] = gamma.inverse().f;
    });
    CPX_HIP(hipMemcpyAsync(d_rgam_.p, h_rgam_.p, B * 2 * sizeof(Fr), hipMemcpyHostToDevice, stream_));
    launch_smsm_round_fold(d_rvec_.p, (int)B, (int)n, (int)half, d_rgam_.p, stream_);
    if (j + 1 == L) {   // x_final (same_multiscalar_argument.rs:138-141)
      CPX_HIP(hipMemcpy2DAsync(h_rfin_.p + 2, 3 * sizeof(Fr), d_rvec_.p, 2 * n * sizeof(Fr), sizeof(Fr), B, hipMemcpyDeviceToHost, stream_));
      wait_stream();
    }
  }

  if (opt_.trace) {
    const ProverState& s = st[0];
    trace_scalar("beta_sp", s.beta_sp.f);
    trace_scalar("alpha_sp", s.alpha_sp.f);
    trace_scalar("-beta_g_inv", (-s.beta_g_inv).f);
    trace_scalar("alpha_g", s.alpha_g.f);
    trace_scalar("gprod", s.gprod.f);
    trace_scalar("beta_g", s.beta_g.f);
    trace_scalar("beta_g_inv", s.beta_g_inv.f);
    trace_scalar("r_p", s.r_p.f);
    trace_scalar("z_ip", s.z_ip.f);
    trace_scalar("alpha_i", s.alpha_i.f);
    trace_scalar("beta_i", s.beta_i.f);
    trace_scalar("alpha_m", s.alpha_m.f);
    trace_scalar("z_k", s.z_k.f);
    trace_scalar("z_t", s.z_t.f);
    trace_scalar("z_u", s.z_u.f);
    trace_scalar("c_final", s.c_final.f);
    trace_scalar("d_final", s.d_final.f);
  }
  // -- serialise
  const size_t psz = proof_size();
  parallel_for(B, [&](size_t p) {
    ProverState& s = st[p];
    uint8_t* o = proofs_out + p * psz;
    auto pt = [&](int slot_id) {
      memcpy(o, &s.comp[(size_t)slot_id * 48], 48);
      o += 48;
    };
    auto sc = [&](const S& x) {
      x.to_le_bytes(o);
      o += 32;
    };
    for (int q = SL_A; q <= SL_C; q++) pt(q);
    sc(s.r_p);
    pt(SL_BC);
    pt(SL_BD);
    for (int q = SL_IPA0; q < SL_IPA0 + 4 * (int)L; q++) pt(q);
    sc(s.c_final);
    sc(s.d_final);
    for (int q = sm.CMA1(); q <= sm.CMB2(); q++) pt(q);
    sc(s.z_k);
    sc(s.z_t);
    sc(s.z_u);
    for (int q = sm.BA(); q < sm.BA() + 3 + 6 * (int)L; q++) pt(q);
    {
      S xf;   // x[0] after the last fold (device-resident vector)
      memcpy(xf.f.v, &h_rfin_.p[3 * p + 2], sizeof(Fr));
      sc(xf);
    }
  });
  flush_timers();
}

// ---------------------------------------------------------------- verifier
void Engine::batch_verify(const uint8_t* proofs, const uint8_t* rand, int* verdict) { verify_core(proofs, rand, 8, verdict, nullptr, nullptr); }
void Engine::batch_verify_fused(const uint8_t* proofs, const uint8_t* rand, uint8_t* partial_jac, int* n_invalid) {
  verify_core(proofs, rand, 12, nullptr, partial_jac, n_invalid);
}

// Shared body.  Per-proof mode (verdict != nullptr, 8 random factors per proof): curdleproofs.rs:197.  Fused mode
// (fused_partial != nullptr, 12 factors per proof): BASELINE config 5 — every check of every proof goes into ONE
// accumulated MSM (the reference's MsmAccumulator shared by all verify calls); the result is this
// engine's partial sum, which must add up to the identity over all engines / GPUs.
void Engine::verify_core(const uint8_t* proofs, const uint8_t* rand, size_t rand_stride, int* verdict, uint8_t* fused_partial, int* fused_invalid) {
  HostSpan wall(this, "host_verify_wall");
  if (!B_) throw std::logic_error("batch_load first");
  CPX_HIP(hipSetDevice(device_));
  const size_t B = B_, ell = ell_, n = n_, L = L_;
  const SlotMap sm(L);
  const size_t psz = proof_size();
  const int NPP = sm.n_proof_points();
  // msm_accumulator.rs:44 draws every factor with Fr::rand; the ABI takes them from the caller, so they are validated:
  // a zero (or non-reduced) factor would silently drop the check it weights
  for (size_t i = 0; i < B * rand_stride; i++)
    if (!host::is_valid_factor(rand + 32 * i)) throw ArgError("verifier random factors must be non-zero reduced field elements");
  {   // infinity encodings as ark-bls12-381 ^0.4 reads them (option strict_infinity = 0): canonical before anything hashes or decodes them
    std::vector<size_t> offs((size_t)NPP);
    for (size_t q = 0; q < (size_t)NPP; q++) offs[q] = 48 * q + (q >= 9 ? 32 : 0) + (q >= 11 + 4 * L ? 64 : 0) + (q >= 15 + 4 * L ? 96 : 0);
    proofs = canonical_infinities(proofs, B * psz, B, psz, offs);
  }
  if (device_prefix(B)) {   // the whole verifier on the GPU (engine_device.cpp); a few proofs: host-driven Fiat-Shamir below
    verify_core_device(proofs, rand, rand_stride, verdict, fused_partial, fused_invalid);
    return;
  }

  TeamScope team(this, B);   // 2 ... device_min_batch - 1 (55) proofs: the host loops on spinning helper threads
  struct VState {
    Transcript tr{"curdleproofs"};
    bool bad = false;           // deserialisation failure
    bool reject = false;        // structural rejection
    S r_p, c_fin, d_fin, z_k, z_t, z_u, x_fin;
    SVec vec_a;
    S alpha_sp, beta_sp, gprod, alpha_g, beta_g, beta_g_inv, z_ip, alpha_i, beta_i, alpha_s, alpha_m;
    SVec gam_i, gam_i_inv, gam_m, gam_m_inv;
    const uint8_t* pb;          // proof bytes
    SVec scal[8];
  };
  std::vector<VState> st(B);

  // byte offset of each proof point (slot order == serialisation order, scalars interleaved)
  std::vector<size_t> pt_off(NPP);
  size_t off_rp, off_c, off_d, off_zk, off_x;
  {
    size_t o = 0;
    int q = 0;
    auto P = [&](int cnt) {
      for (int i = 0; i < cnt; i++) {
        pt_off[q++] = o;
        o += 48;
      }
    };
    P(9);            // A cmT1 cmT2 cmU1 cmU2 R S B C
    off_rp = o;
    o += 32;
    P(2 + 4 * (int)L);
    off_c = o;
    o += 32;
    off_d = o;
    o += 32;
    P(4);
    off_zk = o;
    o += 96;
    P(3 + 6 * (int)L);
    off_x = o;
    o += 32;
    if (o != psz || q != NPP) throw std::logic_error("proof layout");
  }

  // -- V0: compressed instance vectors and M -> affine first (the transcript starts with their bytes: the side stream copies them
  //    to the host), then the proof points are decompressed into their slots — the host hashes (V1a) while that kernel runs
  h_inst_comp_.ensure(B * 4 * ell * 48);
  h_mcomp_.ensure(B * 48);
  const uint8_t* inst_comp = h_inst_comp_.p;
  const uint8_t* mcomp = h_mcomp_.p;
  const size_t npts = (size_t)B * NPP;
  {
    h_pts_.ensure(npts * 48);
    h_u32_.ensure(npts + B);
    h_status_.ensure(npts);
    uint8_t* pts = h_pts_.p;
    uint32_t* dst = h_u32_.p;
    d_bytes_.ensure(B * 4 * ell * 48);
    tick("k_compress", 0, (double)(4 * ell * B));
    launch_compress(d_pp_.p, (int)(4 * ell), (int)pp_stride_, (int)B, d_bytes_.p, stream_);
    tock();
    d_dst_.ensure(B);
    d_comp_.ensure(B * 48);
    uint32_t* mdst = h_u32_.p + npts;
    for (size_t p = 0; p < B; p++) mdst[p] = slot_index(p, SL_M);
    d_mcomp_.ensure(B * 48);
    CPX_HIP(hipMemcpyAsync(d_dst_.p, mdst, B * 4, hipMemcpyHostToDevice, stream_));
    launch_finalize(d_Mjac_.p, (int)B, d_pp_.p, d_dst_.p, d_mcomp_.p, stream_);
    transcript_prefix_async(B);   // side stream: copies of the compressed bytes for the host's transcripts
    parallel_for(B, [&](size_t p) {
      for (int q = 0; q < NPP; q++) {
        memcpy(&pts[(p * NPP + q) * 48], proofs + p * psz + pt_off[q], 48);
        dst[p * NPP + q] = slot_index(p, SL_A + q);
      }
    });
    d_vin_.ensure(npts * 48);
    d_vdst_.ensure(npts);
    d_status_.ensure(npts);
    CPX_HIP(hipMemcpyAsync(d_vin_.p, pts, npts * 48, hipMemcpyHostToDevice, stream_));
    CPX_HIP(hipMemcpyAsync(d_vdst_.p, dst, npts * 4, hipMemcpyHostToDevice, stream_));
    tick("k_decompress", 0, (double)npts);
    launch_decompress(opt_, d_vin_.p, (int)npts, d_pp_.p, d_vdst_.p, d_status_.p, 1, stream_);
    tock();
    CPX_HIP(hipMemcpyAsync(h_status_.p, d_status_.p, npts, hipMemcpyDeviceToHost, stream_));
    wait_side();
  }

  const uint8_t ID0 = 0xc0;
  std::vector<uint8_t> comp;

  // -- V1a: transcript up to the grand-product beta
  parallel_for(B, [&](size_t p) {
    VState& s = st[p];
    s.pb = proofs + p * psz;
    const uint8_t* pb = s.pb;
    auto P = [&](int slot_id) { return pb + pt_off[slot_id - SL_A]; };
    if (!S::from_le_bytes(pb + off_rp, &s.r_p) || !S::from_le_bytes(pb + off_c, &s.c_fin) || !S::from_le_bytes(pb + off_d, &s.d_fin) ||
        !S::from_le_bytes(pb + off_zk, &s.z_k) || !S::from_le_bytes(pb + off_zk + 32, &s.z_t) || !S::from_le_bytes(pb + off_zk + 64, &s.z_u) ||
        !S::from_le_bytes(pb + off_x, &s.x_fin))
      s.bad = true;
    const uint8_t* ic = &inst_comp[p * 4 * ell * 48];
    // curdleproofs.rs:218: the randomiser must not have wiped the ciphertexts
    if (ic[2 * ell * 48] == ID0) s.reject = true;
    for (int v = 0; v < 4; v++) s.tr.append_point_vec_bytes("curdleproofs_step1", ic + v * ell * 48, ell);   // curdleproofs.rs:213-222
    s.tr.append_point_bytes("curdleproofs_step1", &mcomp[p * 48]);
    s.vec_a = s.tr.get_and_append_challenges("curdleproofs_vec_a", ell);
    // same_permutation_argument.rs:131-145
    s.tr.append_point_bytes("same_perm_step1", P(SL_A));
    s.tr.append_point_bytes("same_perm_step1", &mcomp[p * 48]);
    s.tr.append_scalar_vec("same_perm_step1", s.vec_a);
    s.alpha_sp = s.tr.get_and_append_challenge("same_perm_alpha");
    s.beta_sp = s.tr.get_and_append_challenge("same_perm_beta");
    s.gprod = S::one();
    for (size_t i = 0; i < ell; i++) s.gprod *= s.vec_a[i] + S::from_u64(i) * s.alpha_sp + s.beta_sp;
    // grand_product_argument.rs:200-209
    s.tr.append_point_bytes("gprod_step1", P(SL_B));
    s.tr.append_scalar("gprod_step1", s.gprod);
    s.alpha_g = s.tr.get_and_append_challenge("gprod_alpha");
    s.tr.append_point_bytes("gprod_step2", P(SL_C));
    s.tr.append_scalar("gprod_step2", s.r_p);
    s.beta_g = s.tr.get_and_append_challenge("gprod_beta");
    s.beta_g_inv = s.beta_g.inverse();
    s.scal[0] = {S::one(), -s.beta_g_inv, s.alpha_g};    // D  (grand_product_argument.rs:223)
    s.scal[1] = {S::one(), S::one(), S::one()};          // A' (curdleproofs.rs:258)
  });
  {   // the decompressed proof points (and their verdicts) are needed from here on
    wait_stream();
    const uint8_t* status = h_status_.p;
    for (size_t p = 0; p < B; p++)
      for (int q = 0; q < NPP; q++)
        if (status[p * NPP + q]) st[p].bad = true;
  }

  // -- V1b: D = B - beta^-1 sum(G) + alpha sum(H) (grand_product_argument.rs:223) and A' = A + cm_T.T_1 + cm_U.T_1
  //    (curdleproofs.rs:258) are hashed into the transcript, so they are needed as bytes: sums of decompressed proof
  //    points plus two fixed-base terms (G_sum, H_sum are columns of the CRS tables)
  {
    const TblSeg none{nullptr, nullptr, 0, 0};
    std::vector<TblReq> reqs;
    for (size_t p = 0; p < B; p++) {
      TblReq rd{cseg(n + 3, 2), st[p].scal[0].data() + 1, none, nullptr, slot_index(p, sm.D())};
      rd.add[0] = slot_index(p, SL_B);
      TblReq ra{none, nullptr, none, nullptr, slot_index(p, sm.APRIME())};
      ra.add[0] = slot_index(p, SL_A);
      ra.add[1] = slot_index(p, SL_CMT1);
      ra.add[2] = slot_index(p, SL_CMU1);
      reqs.push_back(rd);
      reqs.push_back(ra);
    }
    run_tbl_phase(reqs, &comp);
  }

  // -- V1c: rest of the transcript and the scalars of the accumulated check
  // slot-relative gather list of the "misc" MSM: CRS singles, M, every proof point
  std::vector<uint32_t> misc_idx;
  for (int q = SL_H; q <= SL_M; q++) misc_idx.push_back((uint32_t)q);
  for (int q = 0; q < NPP; q++) misc_idx.push_back((uint32_t)(SL_A + q));
  const size_t NM = misc_idx.size();

  parallel_for(B, [&](size_t p) {
    VState& s = st[p];
    const uint8_t* pb = s.pb;
    auto P = [&](int slot_id) { return pb + pt_off[slot_id - SL_A]; };
    const S a1 = S_from_wire(rand + (p * rand_stride + 0) * 32), a2 = S_from_wire(rand + (p * rand_stride + 1) * 32), a3 = S_from_wire(rand + (p * rand_stride + 2) * 32),
            a4 = S_from_wire(rand + (p * rand_stride + 3) * 32), a5 = S_from_wire(rand + (p * rand_stride + 4) * 32), a6 = S_from_wire(rand + (p * rand_stride + 5) * 32),
            a7 = S_from_wire(rand + (p * rand_stride + 6) * 32), a8 = S_from_wire(rand + (p * rand_stride + 7) * 32);
    const S beta_l = s.beta_g.pow_u64(ell), beta_l1 = beta_l * s.beta_g;
    s.z_ip = s.r_p * beta_l1 + s.gprod * beta_l - S::one();
    // inner_product_argument.rs:283-290, 202-250
    s.tr.append_point_bytes("ipa_step1", P(SL_C));
    s.tr.append_point_bytes("ipa_step1", &comp[(p * 2) * 48]);   // D
    s.tr.append_scalar("ipa_step1", s.z_ip);
    s.tr.append_point_bytes("ipa_step1", P(SL_BC));
    s.tr.append_point_bytes("ipa_step1", P(SL_BD));
    s.alpha_i = s.tr.get_and_append_challenge("ipa_alpha");
    s.beta_i = s.tr.get_and_append_challenge("ipa_beta");
    s.gam_i.resize(L);
    for (size_t j = 0; j < L; j++) {
      s.tr.append_point_bytes("ipa_loop", P(sm.LC((int)j)));
      s.tr.append_point_bytes("ipa_loop", P(sm.LD((int)j)));
      s.tr.append_point_bytes("ipa_loop", P(sm.RC((int)j)));
      s.tr.append_point_bytes("ipa_loop", P(sm.RD((int)j)));
      s.gam_i[j] = s.tr.get_and_append_challenge("ipa_gamma");
    }
    s.gam_i_inv = s.gam_i;
    host::batch_inverse(s.gam_i_inv);
    // same_scalar_argument.rs:112-128
    const int sp[10] = {SL_R, SL_S, SL_CMT1, SL_CMT2, SL_CMU1, SL_CMU2, sm.CMA1(), sm.CMA2(), sm.CMB1(), sm.CMB2()};
    for (int q : sp) s.tr.append_point_bytes("sameexp_points", P(q));
    s.alpha_s = s.tr.get_and_append_challenge("same_scalar_alpha");
    // same_multiscalar_argument.rs:229-233, 167-186
    s.tr.append_point_bytes("same_msm_step1", &comp[(p * 2 + 1) * 48]);   // A'
    s.tr.append_point_bytes("same_msm_step1", P(SL_CMT2));
    s.tr.append_point_bytes("same_msm_step1", P(SL_CMU2));
    {
      const uint8_t* ic = &inst_comp[p * 4 * ell * 48];
      std::vector<uint8_t> vb(n * 48, 0);
      memcpy(vb.data(), ic + 2 * ell * 48, ell * 48);
      for (int i = 0; i < 4; i++) vb[(ell + i) * 48] = ID0;
      memcpy(&vb[(ell + 2) * 48], crs_H_comp_, 48);
      s.tr.append_point_vec_bytes("same_msm_step1", vb.data(), n);
      std::fill(vb.begin() + ell * 48, vb.end(), 0);
      memcpy(vb.data(), ic + 3 * ell * 48, ell * 48);
      for (int i = 0; i < 4; i++) vb[(ell + i) * 48] = ID0;
      memcpy(&vb[(ell + 3) * 48], crs_H_comp_, 48);
      s.tr.append_point_vec_bytes("same_msm_step1", vb.data(), n);
    }
    s.tr.append_point_bytes("same_msm_step1", P(sm.BA()));
    s.tr.append_point_bytes("same_msm_step1", P(sm.BT()));
    s.tr.append_point_bytes("same_msm_step1", P(sm.BU()));
    s.alpha_m = s.tr.get_and_append_challenge("same_msm_alpha");
    s.gam_m.resize(L);
    for (size_t j = 0; j < L; j++) {
      const int six[6] = {sm.LA((int)j), sm.LT((int)j), sm.LU((int)j), sm.RA((int)j), sm.RT((int)j), sm.RU((int)j)};
      for (int q : six) s.tr.append_point_bytes("same_msm_loop", P(q));
      s.gam_m[j] = s.tr.get_and_append_challenge("same_msm_gamma");
    }
    s.gam_m_inv = s.gam_m;
    host::batch_inverse(s.gam_m_inv);

    // verification scalars s_i = prod_{j : bit (L-1-j) of i set} gamma_j  (util.rs:40-64), built by doubling
    auto svec = [&](const SVec& g) {
      SVec sv(n);
      sv[0] = S::one();
      for (size_t j = 0; j < L; j++) {          // after step j, entries < 2^(j+1) are final for the low (j+1) bits
        const size_t w = size_t(1) << j;
        const S gj = g[L - 1 - j];              // bit j of i  <->  round L-1-j
        for (size_t i = 0; i < w; i++) sv[w + i] = sv[i] * gj;
      }
      return sv;
    };
    const SVec s_i = svec(s.gam_i), s_m = svec(s.gam_m);
    SVec s_i_inv = s_i;
    host::batch_inverse(s_i_inv);
    // u (grand_product_argument.rs:211-219)
    SVec u(n);
    {
      S pw = s.beta_g_inv;
      for (size_t i = 0; i < ell; i++) {
        u[i] = pw;
        pw *= s.beta_g_inv;
      }
      for (size_t i = ell; i < n; i++) u[i] = pw;
    }
    // ---- flattened accumulated check: sum_i a_i * (lhs_i - x_i . V_i) == O  (msm_accumulator.rs:38-68) ----
    const S a2c = a2 * s.c_fin, a3d = a3 * s.d_fin, a4x = a4 * s.x_fin, a5x = a5 * s.x_fin, a6x = a6 * s.x_fin;
    // (1) CRS part over G | Hvec (n bases)
    SVec& k1 = s.scal[0];
    k1.assign(n, S::zero());
    const S a1b = a1 * s.beta_sp;
    for (size_t i = 0; i < n; i++) {
      S t = a2c * s_i[i] + a3d * (s_i_inv[i] * u[i]);
      if (i < ell) t += a1b;
      if (i < ell + 2) t += a4x * s_m[i];       // G_b = G | Hvec[0..2) | G_t | G_u
      k1[i] = -t;
    }
    // (2) instance part over R | S | T | U
    SVec& k2 = s.scal[1];
    k2.resize(4 * ell);
    for (size_t i = 0; i < ell; i++) {
      k2[i] = -(a7 * s.vec_a[i]);
      k2[ell + i] = -(a8 * s.vec_a[i]);
      k2[2 * ell + i] = -(a5x * s_m[i]);
      k2[3 * ell + i] = -(a6x * s_m[i]);
    }
    // (3) misc part: singles, M, proof points
    SVec& k3 = s.scal[2];
    k3.assign(NM, S::zero());
    auto at = [&](int slot_id) -> S& { return k3[slot_id <= SL_M ? slot_id : (SL_M + 1) + (slot_id - SL_A)]; };
    at(SL_GT) = -(a4x * s_m[ell + 2]);
    at(SL_GU) = -(a4x * s_m[ell + 3]);
    at(SL_H) = a2 * (s.alpha_i * s.alpha_i * s.z_ip * s.beta_i) - a2c * s.d_fin * s.beta_i - a5x * s_m[ell + 2] - a6x * s_m[ell + 3];
    at(SL_GSUM) = -(a3 * s.alpha_i * s.beta_g_inv);
    at(SL_HSUM) = a3 * s.alpha_i * s.alpha_g;
    at(SL_M) = -(a1 * s.alpha_sp);
    at(SL_B) = a1 + a3 * s.alpha_i;
    at(SL_A) = a4 * s.alpha_m - a1;
    at(SL_CMT1) = a4 * s.alpha_m;
    at(SL_CMU1) = a4 * s.alpha_m;
    at(SL_CMT2) = a5 * s.alpha_m;
    at(SL_CMU2) = a6 * s.alpha_m;
    at(SL_C) = a2 * s.alpha_i;
    at(SL_BC) = a2;
    at(SL_BD) = a3;
    at(sm.BA()) = a4;
    at(sm.BT()) = a5;
    at(sm.BU()) = a6;
    at(SL_R) = a7;
    at(SL_S) = a8;
    for (size_t j = 0; j < L; j++) {
      at(sm.LC((int)j)) = a2 * s.gam_i[j];
      at(sm.RC((int)j)) = a2 * s.gam_i_inv[j];
      at(sm.LD((int)j)) = a3 * s.gam_i[j];
      at(sm.RD((int)j)) = a3 * s.gam_i_inv[j];
      at(sm.LA((int)j)) = a4 * s.gam_m[j];
      at(sm.RA((int)j)) = a4 * s.gam_m_inv[j];
      at(sm.LT((int)j)) = a5 * s.gam_m[j];
      at(sm.RT((int)j)) = a5 * s.gam_m_inv[j];
      at(sm.LU((int)j)) = a6 * s.gam_m[j];
      at(sm.RU((int)j)) = a6 * s.gam_m_inv[j];
    }
    // SameScalar equalities (same_scalar_argument.rs:127-137), each of the form "... == O":
    //   cm_A.T_1 + alpha cm_T.T_1 - z_t G_t,   cm_A.T_2 + alpha cm_T.T_2 - z_k R - z_t H,
    //   cm_B.T_1 + alpha cm_U.T_1 - z_u G_u,   cm_B.T_2 + alpha cm_U.T_2 - z_k S - z_u H
    {
      // The four equalities join the accumulated MSM with random weights of their own: factors 9..12 in fused mode;
      // in per-proof mode the pairwise products a1 a2, a3 a4, a5 a6, a7 a8 of the caller's eight factors — the
      // accumulated sum is then a polynomial of degree 2 in independent uniform factors whose coefficients are the
      // individual check values, so it vanishes with probability <= 2/r unless every check holds (Schwartz-Zippel),
      // the same argument that backs msm_accumulator.rs itself.
      S w1, w2, w3, w4;
      if (fused_partial) {
        w1 = S_from_wire(rand + (p * rand_stride + 8) * 32);
        w2 = S_from_wire(rand + (p * rand_stride + 9) * 32);
        w3 = S_from_wire(rand + (p * rand_stride + 10) * 32);
        w4 = S_from_wire(rand + (p * rand_stride + 11) * 32);
      } else {
        w1 = a1 * a2;
        w2 = a3 * a4;
        w3 = a5 * a6;
        w4 = a7 * a8;
      }
      at(sm.CMA1()) += w1;
      at(SL_CMT1) += w1 * s.alpha_s;
      at(SL_GT) += -(w1 * s.z_t);
      at(sm.CMA2()) += w2;
      at(SL_CMT2) += w2 * s.alpha_s;
      at(SL_R) += -(w2 * s.z_k);
      at(SL_H) += -(w2 * s.z_t);
      at(sm.CMB1()) += w3;
      at(SL_CMU1) += w3 * s.alpha_s;
      at(SL_GU) += -(w3 * s.z_u);
      at(sm.CMB2()) += w4;
      at(SL_CMU2) += w4 * s.alpha_s;
      at(SL_S) += -(w4 * s.z_k);
      at(SL_H) += -(w4 * s.z_u);
    }
  });

  if (fused_partial) {
    // ---- one MSM over the CRS (scalars summed over the proofs, fixed-base table) and B * (4 ell + NM) per-proof points ----
    if (!fixtab()) throw std::logic_error("set_crs first");
    const size_t NI = 4 * ell, NPT = NI + NM, N = B * NPT;
    int invalid = 0;
    for (size_t p = 0; p < B; p++) invalid += (st[p].bad || st[p].reject) ? 1 : 0;
    SVec crs_sum(n, S::zero());
    parallel_for(n, [&](size_t i) {
      S t = S::zero();
      for (size_t p = 0; p < B; p++)
        if (!st[p].bad) t += st[p].scal[0][i];
      crs_sum[i] = t;
    });
    const int fix_wpw = msm_fix_windows_per_wave(opt_, 1, fix_bits_);
    const int fix_parts = msm_fix_parts(fix_bits_, fix_wpw);
    d_scal_.ensure(N + n);
    d_big_idx_.ensure(N);
    d_tpart_.ensure(fix_parts);
    d_ftasks_.ensure(1);
    d_res_.ensure(1);
    const size_t b_scal = (N + n) * sizeof(Fr), b_idx = N * sizeof(uint32_t);
    h_stage_.ensure(b_scal + b_idx + sizeof(FixTask));
    Fr* hs = reinterpret_cast<Fr*>(h_stage_.p);
    uint32_t* hi = reinterpret_cast<uint32_t*>(h_stage_.p + b_scal);
    FixTask* hf = reinterpret_cast<FixTask*>(h_stage_.p + b_scal + b_idx);
    parallel_for(B, [&](size_t p) {
      const VState& s = st[p];
      Fr* d = hs + p * NPT;
      uint32_t* x = hi + p * NPT;
      for (size_t i = 0; i < NI; i++) {
        d[i] = s.bad ? S::zero().f : s.scal[1][i].f;   // an undecodable proof contributes nothing (and is counted as invalid)
        x[i] = (uint32_t)(p * pp_stride_ + i);
      }
      for (size_t j = 0; j < NM; j++) {
        d[NI + j] = s.bad ? S::zero().f : s.scal[2][j].f;
        x[NI + j] = slot_index(p, (int)misc_idx[j]);
      }
    });
    for (size_t i = 0; i < n; i++) hs[N + i] = crs_sum[i].f;
    *hf = FixTask{nullptr, d_scal_.p + N, 0, (uint32_t)n, 0, 0};
    CPX_HIP(hipMemcpyAsync(d_scal_.p, hs, b_scal, hipMemcpyHostToDevice, stream_));
    CPX_HIP(hipMemcpyAsync(d_big_idx_.p, hi, b_idx, hipMemcpyHostToDevice, stream_));
    CPX_HIP(hipMemcpyAsync(d_ftasks_.p, hf, sizeof(FixTask), hipMemcpyHostToDevice, stream_));
    ensure_raw_sets(fix_parts);
    launch_msm_fix(d_ftasks_.p, 1, fixtab(), fix_bits_, fix_wpw, (int)nc(), d_raw_.p, d_rawslot_.p, stream_);
    reduce_sets(fix_parts, 0);
    {
      // The per-proof points of up to 256 groups of proofs through the endomorphism bucket-list kernel (32 additions per
      // point like the per-proof verifier, but one bucket reduction and one Horner tail per GROUP of proofs), then two
      // plain summation levels; the last one adds the fixed-base part and hands the sum over in the standard form.
      const size_t G = (B + 255) / 256, NT = (B + G - 1) / G, NT16 = (NT + 15) / 16 * 16;
      d_tasks_.ensure(NT);
      d_ttasks_.ensure(NT);
      d_conv_.ensure(2 * N);
      d_digits_.ensure(9 * N);
      d_part_.ensure(NT * 32);
      d_wsum_.ensure(NT16 + NT16 / 16);
      ensure_raw_sets(std::max<size_t>(NT * 32, fix_parts));
      std::vector<MsmTask> mt(NT);
      for (size_t g = 0; g < NT; g++) {
        const size_t off = g * G * NPT, np = std::min(G, B - g * G) * NPT;
        mt[g] = MsmTask{d_pp_.p, d_big_idx_.p + off, d_scal_.p + off, (uint32_t)np, 0, (uint32_t)off};
      }
      CPX_HIP(hipMemcpyAsync(d_tasks_.p, mt.data(), NT * sizeof(MsmTask), hipMemcpyHostToDevice, stream_));
      CPX_HIP(hipMemsetAsync(d_wsum_.p, 0, (NT16 + NT16 / 16) * sizeof(TJac), stream_));   // all-zero = identity: pads the summation levels
      tick("k_msm_tblw<2, true>", 128.0 * N, (double)N);
      launch_msm_endo(d_tasks_.p, (int)NT, (int)(G * NPT), d_conv_.p, d_digits_.p, d_ttasks_.p, d_raw_.p, d_rawslot_.p, stream_);
      tock();
      reduce_sets(0, NT * 32, d_part_.p);
      tick("k_msm_tail", 0, (double)NT, true);
      launch_msm_tail(opt_, d_part_.p, d_wsum_.p, nullptr, (int)NT, 16, 8, stream_, nullptr, 0, 2);                      // windows of a group
      launch_msm_tail(opt_, d_wsum_.p, d_wsum_.p + NT16, nullptr, (int)(NT16 / 16), 16, 0, stream_);                     // 16 groups each
      launch_msm_tail(opt_, d_wsum_.p + NT16, nullptr, d_res_.p, 1, (int)(NT16 / 16), 0, stream_, d_tpart_.p, fix_parts);   // + the fixed-base part
      tock();
      CPX_HIP(hipStreamSynchronize(stream_));   // mt (host) was copied asynchronously
    }
    h_comp_.ensure(sizeof(Jac));
    CPX_HIP(hipMemcpyAsync(h_comp_.p, d_res_.p, sizeof(Jac), hipMemcpyDeviceToHost, stream_));
    wait_stream();
    memcpy(fused_partial, h_comp_.p, sizeof(Jac));
    if (fused_invalid) *fused_invalid = invalid;
    flush_timers();
    return;
  }

  // -- V2: the accumulated check of every proof as ONE sum: the CRS part on the fixed-base table (k_msm_fix), all the
  //    per-proof points (R | S | T | U and the slots) in one bucket MSM (k_msm_accw); the Horner tail adds the two.
  {
    if (!fixtab()) throw std::logic_error("set_crs first");
    const size_t NI = 4 * ell, NPT = NI + NM;
    std::vector<uint32_t> all_idx(NPT);   // row-relative gather list: instance vectors, then the slots of the misc part
    for (size_t i = 0; i < NI; i++) all_idx[i] = (uint32_t)i;
    for (size_t j = 0; j < NM; j++) all_idx[NI + j] = (uint32_t)(NI + misc_idx[j]);
    const uint32_t* d_all = idx_list(all_idx);
    const int fix_wpw = msm_fix_windows_per_wave(opt_, (int)B, fix_bits_);
    const int fix_parts = msm_fix_parts(fix_bits_, fix_wpw);
    const size_t slices = (size_t)msm_tblw_slices(opt_, (int)B, 2, (int)NPT);   // a lone proof: several waves per window of the per-proof MSM
    const size_t total = B * (NPT + n);
    d_scal_.ensure(total);
    d_tasks_.ensure(B);
    d_ftasks_.ensure(B);
    d_conv_.ensure(2 * B * NPT);   // points and their endomorphism images
    d_digits_.ensure(9 * B * NPT);
    d_ttasks_.ensure(B);
    d_part_.ensure(B * 32 * slices);
    d_tpart_.ensure(B * (size_t)fix_parts);
    d_res_.ensure(B);
    d_comp_.ensure(B * 48);
    ensure_raw_sets(B * std::max<size_t>(fix_parts, 32 * slices));
    const size_t b_scal = total * sizeof(Fr), b_mt = B * sizeof(MsmTask), b_ft = B * sizeof(FixTask);
    h_stage_.ensure(b_scal + b_mt + b_ft);
    Fr* hs = reinterpret_cast<Fr*>(h_stage_.p);
    MsmTask* hm = reinterpret_cast<MsmTask*>(h_stage_.p + b_scal);
    FixTask* hf = reinterpret_cast<FixTask*>(h_stage_.p + b_scal + b_mt);
    parallel_for(B, [&](size_t p) {
      const VState& s = st[p];
      Fr* d = hs + p * (NPT + n);
      for (size_t i = 0; i < NI; i++) d[i] = s.scal[1][i].f;
      for (size_t j = 0; j < NM; j++) d[NI + j] = s.scal[2][j].f;
      for (size_t i = 0; i < n; i++) d[NPT + i] = s.scal[0][i].f;
      hm[p] = MsmTask{pp(p), d_all, d_scal_.p + p * (NPT + n), (uint32_t)NPT, 0, (uint32_t)(p * NPT)};
      hf[p] = FixTask{nullptr, d_scal_.p + p * (NPT + n) + NPT, 0, (uint32_t)n, 0, (uint32_t)(p * fix_parts)};
    });
    CPX_HIP(hipMemcpyAsync(d_scal_.p, hs, b_scal, hipMemcpyHostToDevice, stream_));
    CPX_HIP(hipMemcpyAsync(d_tasks_.p, hm, b_mt, hipMemcpyHostToDevice, stream_));
    CPX_HIP(hipMemcpyAsync(d_ftasks_.p, hf, b_ft, hipMemcpyHostToDevice, stream_));
    tick(fix_kernel_name(fix_bits_, fix_wpw), 128.0 * n * B,
         (double)(n * B));
    launch_msm_fix(d_ftasks_.p, (int)B, fixtab(), fix_bits_, fix_wpw, (int)nc(), d_raw_.p, d_rawslot_.p, stream_);
    tock();
    reduce_sets(B * fix_parts, 0);
    // per-proof points: used once, so no shifted tables — endomorphism split + radix-256 buckets per window
    tick("k_msm_tblw<2, true>", 128.0 * NPT * B, (double)(NPT * B));
    launch_msm_endo(d_tasks_.p, (int)B, (int)NPT, d_conv_.p, d_digits_.p, d_ttasks_.p, d_raw_.p, d_rawslot_.p, stream_, (int)slices);
    tock();
    reduce_sets(0, B * 32 * slices, d_part_.p);
    tick("k_msm_tail", 0, (double)B);
    launch_msm_tail(opt_, d_part_.p, nullptr, d_res_.p, (int)B, 16, 8, stream_, d_tpart_.p, fix_parts, (int)(2 * slices));
    tock();
    tick("k_finalize", 0, (double)B);
    launch_finalize(d_res_.p, (int)B, nullptr, nullptr, d_comp_.p, stream_);
    tock();
    h_comp_.ensure(B * 48);
    CPX_HIP(hipMemcpyAsync(h_comp_.p, d_comp_.p, B * 48, hipMemcpyDeviceToHost, stream_));
    wait_stream();
    for (size_t p = 0; p < B; p++) {
      const VState& s = st[p];
      const bool ok = h_comp_.p[p * 48] == ID0;
      verdict[p] = s.bad ? CPX_ERR_DESERIALIZE : ((s.reject || !ok) ? CPX_ERR_VERIFY : CPX_OK);
    }
  }
  flush_timers();
}

}  // namespace cpx
