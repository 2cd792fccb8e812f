// The host's persistent worker pool: the verifier front-end (shuffle_verify.cpp), the pooled batch operators and the deferred G1Point
// evaluation (lazy_host.cpp) share ONE set of threads per process.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace cg1 {

// Persistent worker pool (spawning a thread per core per call costs more than a sub-batch of proofs).  Leaked on
// purpose: its threads sleep on a condition variable until the process exits.
class Pool {
 public:
  // The pool of THIS process: threads do not survive fork(), so a forked child (multiprocessing workers that inherited a parent which had
  // already used the pool) starts a pool of its own at its first use -- the parent's object is abandoned there, never touched (its mutexes
  // may have been held by threads that no longer exist).
  static Pool& get() {
    static std::once_flag reg;
    std::call_once(reg, [] { pthread_atfork(nullptr, nullptr, [] { slot().store(nullptr, std::memory_order_release); }); });
    Pool* p = slot().load(std::memory_order_acquire);
    if (!p) {
      Pool* fresh = new Pool;
      if (slot().compare_exchange_strong(p, fresh, std::memory_order_acq_rel)) p = fresh;      // (a loser's pool idles on, unreferenced: rare and harmless)
    }
    return *p;
  }
  size_t size() const { return threads_.size(); }
  // run `job` on `nt` of the pool's threads (the caller's thread also takes part) and wait for all of them
  void run(const std::function<void()>& job, size_t nt) {
    std::lock_guard<std::mutex> serial(run_mutex_);
    if (nt > threads_.size() + 1) nt = threads_.size() + 1;
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &job;
      want_ = nt - 1;
      pending_ = nt - 1;
      ++generation_;
    }
    cv_work_.notify_all();
    job();
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  static std::atomic<Pool*>& slot() { static std::atomic<Pool*> s{nullptr}; return s; }
  // CPUs this process may really use: the affinity mask, cut down to the cgroup CPU quota when there is one (a
  // container that sees 256 cores but owns 16 cores' worth of quota gets throttled, not faster, with 256 threads)
  static size_t usable_cpus() {
    size_t n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (size_t)CPU_COUNT(&set);
    if (n == 0) n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    long long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                         // cgroup v2: "<quota|max> <period>"
      char q[32] = {0};
      if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
      fclose(f);
    } else {
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
    }
    if (quota > 0 && period > 0) {
      size_t q = (size_t)((quota + period - 1) / period);
      if (q >= 1 && q < n) n = q;
    }
    return n;
  }
  Pool() {
    size_t n = usable_cpus();
    if (const char* e = getenv("CURDLE_G1_THREADS")) { long v = atol(e); if (v >= 1) n = (size_t)v; }
    if (n > 256) n = 256;
    for (size_t i = 0; i + 1 < n; ++i) threads_.emplace_back([this, i] { loop(i); });
    for (auto& t : threads_) t.detach();
  }
  void loop(size_t index) {
    size_t seen = 0;
    for (;;) {
      const std::function<void()>* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (index < want_) job = job_;
      }
      if (job) {
        (*job)();
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) cv_done_.notify_all();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_, run_mutex_;
  std::condition_variable cv_work_, cv_done_;
  const std::function<void()>* job_ = nullptr;
  size_t want_ = 0, pending_ = 0, generation_ = 0;
};

}  // namespace cg1
