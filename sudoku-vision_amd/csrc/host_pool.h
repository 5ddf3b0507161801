// Persistent host worker pool shared by the host-side translation units (corner search, JPEG entropy decoding).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <thread>
#include <vector>

// A small persistent worker pool: batch calls arrive every millisecond or so in the streaming pipeline, and spawning
// 16 std::threads per call cost about as much as the search itself.  parallel_for(n, threads, fn) runs fn(i) for
// i in [0,n) on up to `threads` workers (dynamic index hand-out) and returns when all are done.
class WorkerPool {
  public:
    static WorkerPool &instance() { static WorkerPool p; return p; }

    void parallel_for(int n, int threads, const std::function<void(int)> &fn)
    {
        if (threads <= 1 || n <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
        std::unique_lock<std::mutex> call_lock(call_mu_);          // one batch at a time
        ensure(threads - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_.store(0); active_ = std::min(threads - 1, (int)workers_.size()); pending_ = active_; gen_++;
        }
        cv_.notify_all();
        drain();                                                    // the calling thread works too
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

    // Restrict the workers (present and future) to these CPUs; n == 0 lifts the restriction for future workers only.  The caller of
    // parallel_for works too and keeps its own mask.
    void set_affinity(const int *cpus, int n)
    {
        std::lock_guard<std::mutex> call_lock(call_mu_);
        CPU_ZERO(&cpus_);
        have_cpus_ = n > 0;
        for (int i = 0; i < n; i++)
            if (cpus[i] >= 0 && cpus[i] < CPU_SETSIZE) CPU_SET(cpus[i], &cpus_);
        if (have_cpus_)
            for (auto &t : workers_) pthread_setaffinity_np(t.native_handle(), sizeof(cpus_), &cpus_);
    }

  private:
    WorkerPool() { CPU_ZERO(&cpus_); }
    ~WorkerPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void ensure(int k)
    {
        while ((int)workers_.size() < k) {
            const int id = (int)workers_.size();
            workers_.emplace_back([this, id] { loop(id); });
            if (have_cpus_) pthread_setaffinity_np(workers_.back().native_handle(), sizeof(cpus_), &cpus_);
        }
    }
    void drain()
    {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
        }
    }
    void loop(int id)
    {
        unsigned long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (stop_) return;
            if (id >= active_) continue;
            lk.unlock();
            drain();
            lk.lock();
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
    std::mutex call_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    const std::function<void(int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, active_ = 0, pending_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
    cpu_set_t cpus_;
    bool have_cpus_ = false;
};
