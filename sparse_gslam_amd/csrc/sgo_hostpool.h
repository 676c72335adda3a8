// sgo_hostpool.h -- host worker pool of the structure builds (no HIP dependency: tests/cpp/hostpool_tsan.cpp
// compiles it alone under -fsanitize=thread).
#pragma once
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sgo {

// Host worker pool for the structure builds (row plan, slot lists, multigrid pattern / product lists): up to
// 32 threads (env SGO_HOST_THREADS; the cores this process may run on), created once per process and parked on a
// condition variable between parallel regions -- spawning threads per region cost more than the regions of a
// 100-ms set-up.  One region at a time (contexts on different threads queue on the pool's mutex).
//
// Region hand-over: a region is {task, ntasks, generation}, published under mu_; a worker copies it under mu_ and
// claims task indices from claim_, whose high word is the generation it belongs to -- a worker that wakes late (or
// still holds an index of the region before) cannot claim from, run or count down a region it has no snapshot of.
// run() returns when every task of its region has FINISHED (done_ == ntasks, counted after the task body), so the
// caller's lambda outlives all of its executions.
class HostPool {
 public:
  // Two pools: lane 0 serves the calling threads, lane 1 the helper thread of the set-up pipeline (build_structure),
  // whose long regions -- the multigrid's symbolic phase -- would otherwise queue behind (and hold up) the many short
  // regions of the structure build.  lane() selects per thread.
  static int& lane() {
    static thread_local int l = 0;
    return l;
  }
  static HostPool& get() { return lane() ? pool1() : pool0(); }
  int size() const { return nthreads_; }
  // fn(t) for t in [0, ntasks), distributed over the workers and the caller
  template <class F>
  void run(int ntasks, F&& fn) {
    if (ntasks <= 0) return;
    if (ntasks == 1 || nthreads_ <= 1) {
      for (int t = 0; t < ntasks; ++t) fn(t);
      return;
    }
    std::lock_guard<std::mutex> region(region_mu_);
    const std::function<void(int)> task = [&fn](int t) { fn(t); };
    uint32_t gen;
    {
      std::lock_guard<std::mutex> lk(mu_);
      gen = ++generation_;
      task_ = &task;
      ntasks_ = ntasks;
      done_.store(0, std::memory_order_relaxed);
      claim_.store((uint64_t)gen << 32, std::memory_order_release);
    }
    cv_.notify_all();
    work(&task, ntasks, gen);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return done_.load(std::memory_order_acquire) == ntasks; });
    task_ = nullptr;
  }

  explicit HostPool(int nthreads) { start(nthreads); }   // tests
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  HostPool(const HostPool&) = delete;
  HostPool& operator=(const HostPool&) = delete;

 private:
  // each lane is created by its first user (a process that never runs the set-up pipeline parks no second pool)
  static HostPool& pool0() {
    static HostPool p;
    return p;
  }
  static HostPool& pool1() {
    static HostPool p;
    return p;
  }
  // Default size: the cores this process may run on, at most 32 -- shared among the rank processes of one node
  // (LOCAL_WORLD_SIZE, as torch.distributed.run exports it: 8 ranks on a node would otherwise park 8 x 2 x 31 threads and
  // oversubscribe the host during the parallel structure build).  SGO_HOST_THREADS overrides.
  HostPool() {
    int n = (int)std::thread::hardware_concurrency();
#if defined(__linux__)
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
#endif
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) n /= std::max(1, std::atoi(e));
    n = std::max(1, std::min(n, 32));
    if (const char* e = std::getenv("SGO_HOST_THREADS")) n = std::max(1, std::atoi(e));
    start(n);
  }
  void start(int n) {
    nthreads_ = std::max(1, n);
    for (int i = 1; i < nthreads_; ++i) workers_.emplace_back([this] { loop(); });
  }
  // Tasks of region `gen` only: the claim fails as soon as claim_ carries another generation.
  void work(const std::function<void(int)>* task, int ntasks, uint32_t gen) {
    for (;;) {
      uint64_t cur = claim_.load(std::memory_order_acquire);
      int t;
      for (;;) {
        if ((uint32_t)(cur >> 32) != gen) return;
        t = (int)(uint32_t)cur;
        if (t >= ntasks) return;
        if (claim_.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel, std::memory_order_acquire)) break;
      }
      (*task)(t);
      if (done_.fetch_add(1, std::memory_order_acq_rel) + 1 == ntasks) {
        std::lock_guard<std::mutex> lk(mu_);
        done_cv_.notify_all();
      }
    }
  }
  void loop() {
    uint32_t seen = 0;
    for (;;) {
      const std::function<void(int)>* task;
      int ntasks;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return quit_ || generation_ != seen; });
        if (quit_) return;
        seen = generation_;
        task = task_;
        ntasks = ntasks_;
      }
      if (task) work(task, ntasks, seen);
    }
  }
  int nthreads_ = 1;
  std::vector<std::thread> workers_;
  std::mutex region_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* task_ = nullptr;   // guarded by mu_
  int ntasks_ = 0;                                   // guarded by mu_
  uint32_t generation_ = 0;                          // guarded by mu_
  std::atomic<uint64_t> claim_{0};                   // (generation << 32) | next task index
  std::atomic<int> done_{0};                         // finished tasks of the current region
  bool quit_ = false;
};

// Static-partition parallel loop over [0, n) on the host pool (structure builds only).
template <class F>
inline void host_parallel_for(int n, int grain, F&& fn) {
  const int T = std::max(1, std::min(HostPool::get().size(), n / std::max(1, grain)));
  if (T == 1) {
    fn(0, n, 0);
    return;
  }
  HostPool::get().run(T, [&](int t) {
    const int lo = (int)((long long)n * t / T), hi = (int)((long long)n * (t + 1) / T);
    fn(lo, hi, t);
  });
}

}  // namespace sgo
