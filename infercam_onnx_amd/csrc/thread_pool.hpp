// thread_pool.hpp -- fixed pool of host workers for the entropy-decode stage (one JPEG per task).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ufd {

class ThreadPool {
 public:
  explicit ThreadPool(unsigned n) {
    for (unsigned i = 1; i < n; i++) workers_.emplace_back([this] { worker(); });
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
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      total_ = n;
      next_.store(0);
      pending_ = n;
      gen_++;
    }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(mu_);
    // also wait until every worker has left drain(): none may straddle two generations
    done_cv_.wait(lk, [this] { return pending_ == 0 && active_ == 0; });
    fn_ = nullptr;
  }

 private:
  void drain() {
    for (;;) {
      unsigned i = next_.fetch_add(1);
      if (i >= total_) return;
      (*fn_)(i);
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_cv_.notify_all();
    }
  }
  void worker() {
    unsigned seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        active_++;
      }
      drain();
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--active_ == 0) done_cv_.notify_all();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(unsigned)>* fn_ = nullptr;
  std::atomic<unsigned> next_{0};
  unsigned total_ = 0, pending_ = 0, gen_ = 0, active_ = 0;
  bool stop_ = false;
};

}  // namespace ufd
