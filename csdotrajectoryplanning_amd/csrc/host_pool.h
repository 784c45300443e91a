// The library's host threads: one lazily started pool for everything the DO phase does on the CPU in front of and behind a launch
// (host bridge, packing, result scatter).  A streamed DO phase calls these stages once per chunk, a few hundred microseconds of work
// each: threads created per call (round 5: std::thread in parallel_for) cost as much as the work of a small first chunk.
// Pure C++, no HIP: also used by the lane-serial test build.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/csdo_dsqp.h"

namespace csdo {

class HostPool {
 public:
  // one loop over [0, n): items are claimed from `next`; `left` counts the items not yet finished
  struct Job {
    std::atomic<int> next{0}, left{0}, failed{0};
    int n = 0;
    void (*call)(void*, int) = nullptr;   // body(i) of the caller's frame: touched only for a claimed i < n, i.e. while left > 0
    void* frame = nullptr;
    void work() {
      for (;;) {
        const int i = next.fetch_add(1, std::memory_order_relaxed);
        if (i >= n) break;
        try {
          call(frame, i);
        } catch (...) {
          failed.store(1, std::memory_order_relaxed);
        }
        left.fetch_sub(1, std::memory_order_release);
      }
    }
  };

  static HostPool& get() {
    static HostPool* p = new HostPool();   // never destroyed: its threads sleep until the process ends (no join at exit, no
    return *p;                             // unload order to get wrong; a forked child finds no threads and runs its loops alone)
  }
  int helpers() const { return (int)thr_.size(); }

  // tickets: how many pool threads may join the job (the caller works too).  Sleeping threads are woken LAST IN, FIRST OUT: the thread
  // that went to sleep most recently has the warmest core (a condition variable shared by all would wake the one that has slept
  // longest - on a 256-core host that is a core in its deepest idle state: one world's bridge took 3.4 ms instead of 0.4 after a
  // 50 ms pause).
  void offer(const std::shared_ptr<Job>& job, int tickets) {
    tickets = std::min(tickets, helpers());
    if (tickets < 1) return;
    Worker* wake[64];
    int n_wake = 0;
    {
      std::lock_guard<std::mutex> g(m_);
      for (int k = 0; k < tickets; ++k) q_.push_back(job);
      queued_.fetch_add(tickets, std::memory_order_relaxed);
      // (threads that are still spinning after their last job find the tickets by themselves)
      int need = tickets - spinning_.load(std::memory_order_relaxed);
      while (need-- > 0 && !idle_.empty() && n_wake < 64) {
        wake[n_wake++] = idle_.back();
        idle_.pop_back();
      }
      for (int k = 0; k < n_wake; ++k) wake[k]->go = true;
    }
    for (int k = 0; k < n_wake; ++k) wake[k]->cv.notify_one();
  }
  // the job is done: tickets nobody has taken up are withdrawn (a thread woken for a later job would find them first)
  void retire(const std::shared_ptr<Job>& job) {
    if (queued_.load(std::memory_order_relaxed) < 1) return;
    std::lock_guard<std::mutex> g(m_);
    for (auto it = q_.begin(); it != q_.end();) {
      if (it->get() == job.get()) {
        it = q_.erase(it);
        queued_.fetch_sub(1, std::memory_order_relaxed);
      } else {
        ++it;
      }
    }
  }

 private:
  struct Worker {
    std::condition_variable cv;
    bool go = false;   // under m_
  };
  HostPool() {
    // CSDO_HOST_THREADS caps the pool (1: no pool threads, every loop runs on its caller); default: the cores there are, at most 64
    int want = (int)std::thread::hardware_concurrency() - 1;
    if (const char* e = std::getenv("CSDO_HOST_THREADS")) want = std::atoi(e) - 1;
    want = std::max(0, std::min(want, 63));
    try {
      workers_.reserve((size_t)want);
      for (int k = 0; k < want; ++k) {
        workers_.emplace_back(new Worker());
        Worker* w = workers_.back().get();
        thr_.emplace_back([this, w]() { serve(w); });
      }
    } catch (...) {   // std::system_error (the caller's cgroup may cap threads): the pool is what could be started
    }
    for (auto& t : thr_) t.detach();
    // fork(): only the forking thread exists in the child.  The pool's lock is taken around the fork, so that the child does not inherit
    // it locked by a thread that is not there, and the child starts with an empty pool (every loop runs on its caller).
    pthread_atfork([]() { get().m_.lock(); }, []() { get().m_.unlock(); },
                   []() {
                     HostPool& p = get();
                     p.q_.clear();
                     p.idle_.clear();
                     p.thr_.clear();
                     p.queued_.store(0);
                     p.spinning_.store(0);
                     p.m_.unlock();
                   });
  }
  // A thread that has just worked stays awake for a moment: the stages of a DO phase follow each other within microseconds (a world's
  // bridge is four loops in a row), and waking a sleeping thread costs more than such a loop's share of the work.
  void serve(Worker* me) {
    using clock = std::chrono::steady_clock;
    bool worked = false;
    for (;;) {
      std::shared_ptr<Job> job;
      if (worked) {
        spinning_.fetch_add(1, std::memory_order_relaxed);
        const auto until = clock::now() + std::chrono::microseconds(200);
        do {
          if (queued_.load(std::memory_order_relaxed) > 0) {
            std::lock_guard<std::mutex> g(m_);
            if (!q_.empty()) {
              job = std::move(q_.front());
              q_.pop_front();
              queued_.fetch_sub(1, std::memory_order_relaxed);
              break;
            }
          }
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        } while (clock::now() < until);
        spinning_.fetch_sub(1, std::memory_order_relaxed);
      }
      if (!job) {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
          if (!q_.empty()) {
            job = std::move(q_.front());
            q_.pop_front();
            queued_.fetch_sub(1, std::memory_order_relaxed);
            break;
          }
          idle_.push_back(me);
          me->cv.wait(g, [&]() { return me->go; });
          me->go = false;
        }
      }
      job->work();
      worked = true;
    }
  }
  std::atomic<int> queued_{0}, spinning_{0};
  std::mutex m_;
  std::deque<std::shared_ptr<Job>> q_;
  std::vector<Worker*> idle_;   // sleeping threads, most recent last
  std::vector<std::unique_ptr<Worker>> workers_;
  std::vector<std::thread> thr_;
};

// body(i) for i in [0, n) on the calling thread and up to max_threads - 1 pool threads.  Nothing escapes: a body that throws
// (std::bad_alloc in a growing vector) is recorded and the loop goes on; a pool without threads (thread cap, forked child,
// CSDO_HOST_THREADS=1) leaves everything to the caller.  Nested calls are fine: the caller of every loop works on it itself.
// Returns CSDO_OK, or CSDO_ENOMEM if a body threw.
template <class F>
inline int parallel_for(const int n, const int max_threads, F&& body) {
  if (n < 1) return CSDO_OK;
  using Fn = std::remove_reference_t<F>;
  auto job = std::make_shared<HostPool::Job>();
  job->n = n;
  job->left.store(n, std::memory_order_relaxed);
  job->frame = (void*)&body;
  job->call = [](void* f, int i) { (*(Fn*)f)(i); };
  const bool shared = n > 1 && max_threads > 1;
  if (shared) HostPool::get().offer(job, std::min(n, max_threads) - 1);
  job->work();
  while (job->left.load(std::memory_order_acquire) > 0) std::this_thread::yield();   // items still running on pool threads
  if (shared) HostPool::get().retire(job);
  return job->failed.load() ? CSDO_ENOMEM : CSDO_OK;
}

}  // namespace csdo
