// Host-side thread helpers of the engine — product code, plain C++ (no HIP): the sleeping worker pool of the large host loops and
// the spin team of a small batch's call.  Header-only so that the CPU tests exercise the very code (tests/host_emul/emul.cpp).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>
#include <cstdio>
#include <sched.h>

namespace cpx {

// Persistent host worker pool: the per-phase Fiat-Shamir / Fr work of a batch is spread over these threads
// (spawning threads per phase cost ~100 ms per step at 50 phases x 64 threads).
class WorkerPool {
 public:
  explicit WorkerPool(size_t threads);
  ~WorkerPool();
  size_t size() const { return threads_.size() + 1; }
  // runs f(i) for i in [0, n); the calling thread participates; exceptions are rethrown
  void run(size_t n, const std::function<void(size_t)>& f);

 private:
  void worker();
  void drain();
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(size_t)>* job_ = nullptr;
  size_t n_ = 0, chunk_ = 1, active_ = 0;
  std::atomic<size_t> next_{0};
  uint64_t epoch_ = 0;
  bool stop_ = false;
  std::exception_ptr err_;
};

// A few helper threads that SPIN between jobs while engaged.  The host-driven path of a small batch (2 ... 95 proofs: below the `device_min_batch` option) runs a loop
// over the proofs between every two MSM phases — 20-100 us of transcript and scalar work per proof, every ~0.3 ms, 40 times per
// call: a sleeping pool wakes too slowly for that (tens of us per wake-up) and the loop run inline makes the call scale with the
// batch (12.7 ms at 4 proofs against 6.8 ms at 1).  engage() at the start of the call, release() at its end; in between the
// helpers poll an epoch counter.
// Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (cpu.max; containers on the MI355X hosts
// see 256 logical CPUs and get 16).  Read once.
inline size_t effective_host_cores();

class SpinTeam {
 public:
  explicit SpinTeam(size_t helpers);
  ~SpinTeam();
  size_t size() const { return threads_.size() + 1; }
  void engage();
  void release();
  bool engaged() const { return spinning_.load(std::memory_order_relaxed); }
  void run(size_t n, const std::function<void(size_t)>& f);   // the caller takes part; exceptions are rethrown

 private:
  void worker();
  void drain();
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_ = false;                       // guarded by mu_
  std::atomic<bool> spinning_{false};
  std::atomic<uint64_t> epoch_{0};
  std::atomic<size_t> next_{0}, done_{0}, n_{0}, in_drain_{0};
  std::atomic<const std::function<void(size_t)>*> job_{nullptr};
  std::mutex err_mu_;
  std::exception_ptr err_;
};

inline size_t effective_host_cores() {
  static const size_t cores = [] {
    size_t c = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) c = std::min<size_t>(c, (size_t)CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
      long long quota = 0, period = 0;
      if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) c = std::min<size_t>(c, (size_t)std::max<long long>(1, quota / period));
      fclose(f);
    }
    return c;
  }();
  return cores;
}

inline WorkerPool::WorkerPool(size_t threads) {
  for (size_t i = 1; i < threads; i++) threads_.emplace_back([this] { worker(); });
}
inline WorkerPool::~WorkerPool() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
  }
  cv_.notify_all();
  for (auto& t : threads_) t.join();
}
// Items are claimed through an atomic counter in small chunks (no lock on the hot path); the mutex only
// guards job hand-off and completion.
inline void WorkerPool::drain() {
  const size_t n = n_, chunk = chunk_;
  for (;;) {
    const size_t b = next_.fetch_add(chunk, std::memory_order_relaxed);
    if (b >= n) break;
    const size_t e = std::min(n, b + chunk);
    try {
      for (size_t i = b; i < e; i++) (*job_)(i);
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu_);
      if (!err_) err_ = std::current_exception();
      next_.store(n, std::memory_order_relaxed);
    }
  }
}
inline void WorkerPool::worker() {
  uint64_t seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return stop_ || (epoch_ != seen && job_); });
      if (stop_) return;
      seen = epoch_;
      active_++;
    }
    drain();
    {
      std::lock_guard<std::mutex> lk(mu_);
      active_--;
      if (active_ == 0) done_cv_.notify_all();
    }
  }
}
inline void WorkerPool::run(size_t n, const std::function<void(size_t)>& f) {
  if (!n) return;
  {
    std::lock_guard<std::mutex> lk(mu_);
    job_ = &f;
    n_ = n;
    chunk_ = std::max<size_t>(1, n / (8 * size()));
    next_.store(0, std::memory_order_relaxed);
    err_ = nullptr;
    epoch_++;
    active_++;
  }
  cv_.notify_all();
  drain();
  std::unique_lock<std::mutex> lk(mu_);
  active_--;
  done_cv_.wait(lk, [&] { return active_ == 0; });
  job_ = nullptr;
  if (err_) std::rethrow_exception(err_);
}

inline SpinTeam::SpinTeam(size_t helpers) {
  for (size_t i = 0; i < helpers; i++) threads_.emplace_back([this] { worker(); });
}
inline SpinTeam::~SpinTeam() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
    spinning_.store(false, std::memory_order_release);
  }
  cv_.notify_all();
  for (auto& t : threads_) t.join();
}
inline void SpinTeam::engage() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    spinning_.store(true, std::memory_order_release);
  }
  cv_.notify_all();
}
inline void SpinTeam::release() { spinning_.store(false, std::memory_order_release); }
inline void SpinTeam::drain() {
  const std::function<void(size_t)>* job = job_.load(std::memory_order_acquire);
  const size_t n = n_.load(std::memory_order_acquire);
  if (!job) return;
  for (;;) {
    const size_t i = next_.fetch_add(1, std::memory_order_acq_rel);
    if (i >= n) break;
    try {
      (*job)(i);
    } catch (...) {
      std::lock_guard<std::mutex> lk(err_mu_);
      if (!err_) err_ = std::current_exception();
    }
    done_.fetch_add(1, std::memory_order_acq_rel);
  }
}
inline void SpinTeam::worker() {
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return stop_ || spinning_.load(std::memory_order_acquire); });
      if (stop_) return;
    }
    uint64_t seen = 0;   // (a job of an earlier engagement is complete: draining it again finds no items)
    while (spinning_.load(std::memory_order_acquire)) {
      const uint64_t e = epoch_.load(std::memory_order_acquire);
      if (e != seen) {
        seen = e;
        in_drain_.fetch_add(1, std::memory_order_acq_rel);
        if (epoch_.load(std::memory_order_acquire) == e) drain();   // (the job may have completed and gone in between)
        in_drain_.fetch_sub(1, std::memory_order_acq_rel);
      } else {
        __builtin_ia32_pause();
      }
    }
  }
}
inline void SpinTeam::run(size_t n, const std::function<void(size_t)>& f) {
  if (!n) return;
  err_ = nullptr;
  n_.store(n, std::memory_order_relaxed);
  done_.store(0, std::memory_order_relaxed);
  next_.store(0, std::memory_order_relaxed);
  job_.store(&f, std::memory_order_release);
  epoch_.fetch_add(1, std::memory_order_acq_rel);
  drain();
  while (done_.load(std::memory_order_acquire) < n) __builtin_ia32_pause();
  // no helper may still be inside this job's drain() when the next job resets the counters (or f goes out of scope)
  job_.store(nullptr, std::memory_order_release);
  epoch_.fetch_add(1, std::memory_order_acq_rel);
  while (in_drain_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
  if (err_) std::rethrow_exception(err_);
}

}  // namespace cpx
