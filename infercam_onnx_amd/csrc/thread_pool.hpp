// thread_pool.hpp -- fixed pool of host workers for the entropy-decode stage (one JPEG per task).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ufd {

class ThreadPool {
 public:
  // on_start runs first in every worker thread (CPU affinity of the handle's NUMA node).
  explicit ThreadPool(unsigned n, std::function<void()> on_start = nullptr) {
    for (unsigned i = 1; i < n; i++)
      workers_.emplace_back([this, on_start] {
        if (on_start) on_start();
        worker();
      });
  }
  ~ThreadPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  ThreadPool(const ThreadPool&) = delete;
  ThreadPool& operator=(const ThreadPool&) = delete;

  // Runs fn(i) for i in [0, n); the calling thread participates.  Not re-entrant.
  void parallel_for(unsigned n, const std::function<void(unsigned)>& fn) {
    if (n == 0) return;
    if (workers_.empty() || n == 1) {
      for (unsigned i = 0; i < n; i++) fn(i);
      return;
    }
    Job job;
    {
      std::lock_guard<std::mutex> lk(mu_);
      gen_++;
      job = job_ = Job{&fn, n, gen_};
      pending_ = n;
      // generation and next index live in ONE atomic: a worker that wakes late for an older
      // generation can never claim an index of this one
      cursor_.store((uint64_t)gen_ << 32);
    }
    cv_.notify_all();
    drain(job);
    std::unique_lock<std::mutex> lk(mu_);
    // also wait until every worker has left drain(): `fn` dies with this frame
    done_cv_.wait(lk, [this] { return pending_ == 0 && active_ == 0; });
  }

 private:
  struct Job {
    const std::function<void(unsigned)>* fn = nullptr;
    unsigned total = 0, gen = 0;
  };
  void drain(const Job& job) {
    for (;;) {
      uint64_t cur = cursor_.load();
      unsigned i;
      for (;;) {
        if ((unsigned)(cur >> 32) != job.gen) return;  // a newer generation owns the cursor
        i = (unsigned)cur;
        if (i >= job.total) return;
        if (cursor_.compare_exchange_weak(cur, cur + 1)) break;
      }
      (*job.fn)(i);
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_cv_.notify_all();
    }
  }
  void worker() {
    unsigned seen = 0;
    for (;;) {
      Job job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        job = job_;  // {fn, total, gen} of one generation, read under the lock that published them
        active_++;
      }
      drain(job);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--active_ == 0) done_cv_.notify_all();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  Job job_;
  std::atomic<uint64_t> cursor_{0};
  unsigned pending_ = 0, gen_ = 0, active_ = 0;
  bool stop_ = false;
};

}  // namespace ufd
