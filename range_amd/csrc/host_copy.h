// Host side of the reference's output contract: `model(x)` returns a FRESH host ndarray
// (B,1280) float64 (range/range.py:240).  A fresh 100 MB array is 25 000 untouched pages: one
// thread filling it spends ~12 ms in first-touch page faults, more than half of the GPU time of
// the batch.  HostCopyPool spreads the copy (and with it the faults) over a few threads, slab by
// slab as the slabs land in pinned staging memory, while the device -> host DMA of the next slab
// is in flight.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace range_host {

// A fixed set of worker threads running one job at a time: job(t, n_threads) on every worker t.
class HostCopyPool {
   public:
    explicit HostCopyPool(int n_threads) : n_(n_threads < 1 ? 1 : n_threads) {
        for (int t = 0; t < n_; ++t) workers_.emplace_back([this, t] { loop(t); });
    }
    ~HostCopyPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    int size() const { return n_; }

    // runs job(t, size()) on all workers and returns when every one has finished.  One job at a
    // time: callers from several threads (ctypes drops the GIL around the library's calls) queue
    // up on call_mu_ - without it a second caller would overwrite job_ / pending_ / generation_
    // under the first one's wait (a partially filled array, or a deadlock).
    void run(const std::function<void(int, int)>& job) {
        std::lock_guard<std::mutex> one_caller(call_mu_);
        std::unique_lock<std::mutex> g(mu_);
        job_ = &job;
        pending_ = n_;
        ++generation_;
        cv_.notify_all();
        done_cv_.wait(g, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

    // dst[0, bytes) = src[0, bytes): contiguous shares, 4 KB aligned so that no page is faulted by
    // two threads
    void copy(void* dst, const void* src, size_t bytes) {
        if (bytes < (size_t)1 << 20) {
            std::memcpy(dst, src, bytes);
            return;
        }
        run([=](int t, int n) {
            const size_t per = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
            const size_t lo = per * (size_t)t;
            if (lo >= bytes) return;
            const size_t len = lo + per <= bytes ? per : bytes - lo;
            std::memcpy((char*)dst + lo, (const char*)src + lo, len);
        });
    }

    static int default_threads() {
        if (const char* e = std::getenv("RANGE_HOST_THREADS")) {
            const int v = std::atoi(e);
            if (v > 0) return v < 64 ? v : 64;
        }
        const unsigned hc = std::thread::hardware_concurrency();
        const int v = hc ? (int)hc : 8;
        return v < 8 ? v : 8;   // (8, 16 and 32 threads measured alike; MADV_POPULATE_WRITE prefaulting did not help)
    }

   private:
    void loop(int t) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)>* job;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                job = job_;
            }
            (*job)(t, n_);
            {
                std::lock_guard<std::mutex> g(mu_);
                if (--pending_ == 0) done_cv_.notify_all();
            }
        }
    }
    const int n_;
    std::vector<std::thread> workers_;
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int, int)>* job_ = nullptr;
    uint64_t generation_ = 0;
    int pending_ = 0;
    bool stop_ = false;
};

}  // namespace range_host
