/* worker_pool.hpp — a few long-lived host threads shared by the drop-in headers (the chunks of a search round,
 * the cloud's content hash). Host only; no dependency on the library. */
#ifndef TRIPLET_MATCH_WORKER_POOL_HPP_
#define TRIPLET_MATCH_WORKER_POOL_HPP_

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace triplet_match {
namespace detail {

/* The reference starts hardware_concurrency() - 1 std::async tasks per find_parallel call
 * (include/impl/scene.hpp:146-166); a thread per chunk and round cost more than the 1.6 ms of sampling it spread
 * out. run(n, fn) calls fn(0 .. n-1), the caller taking part, and returns when all are done; one batch at a time.
 * The threads are created on first use and joined when the process exits.
 * A task that throws (on the caller's thread or on a worker) does not tear anything down: the first exception is
 * kept, the indices not yet started are passed over, run() waits until every thread has left the batch and rethrows
 * on the caller — as the reference's fut.get() would (:168-171). run() from inside a task (a pooled task that hashes
 * a cloud, say) runs inline on that thread instead of waiting for the pool it is part of. */
class worker_pool {
public:
    static worker_pool& instance() {
        static worker_pool pool;
        return pool;
    }
    size_t size() const { return threads_.size(); }
    void run(size_t n, const std::function<void(size_t)>& fn) {
        if (n <= 1 || threads_.empty() || in_task()) {
            for (size_t i = 0; i < n; ++i) fn(i);
            return;
        }
        std::lock_guard<std::mutex> one_batch(run_mutex_);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            total_ = n;
            next_.store(0);
            done_ = 0;
            failed_.store(false);
            error_ = nullptr;
            ++generation_;
        }
        cv_.notify_all();
        work(fn, n);
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [&] { return done_ == total_ && active_ == 0; });
            job_ = nullptr;
            err = error_;
            error_ = nullptr;
        }
        if (err) std::rethrow_exception(err);
    }

private:
    worker_pool() {
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned n = hw > 1 ? std::min(6u, hw - 1u) : 0u;
        for (unsigned t = 0; t < n; ++t) threads_.emplace_back([this] { loop(); });
    }
    ~worker_pool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    static bool& in_task() {
        static thread_local bool flag = false;
        return flag;
    }
    void work(const std::function<void(size_t)>& fn, size_t n) {
        size_t finished = 0;
        std::exception_ptr err;
        bool& nested = in_task();
        nested = true;
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= n) break;
            if (!failed_.load(std::memory_order_relaxed)) {
                try {
                    fn(i);
                } catch (...) {
                    if (!err) err = std::current_exception();
                    failed_.store(true);
                }
            }
            ++finished; /* an index passed over after a failure still counts: run() waits for done_ == total_ */
        }
        nested = false;
        if (finished || err) {
            std::lock_guard<std::mutex> lk(m_);
            done_ += finished;
            if (err && !error_) error_ = err;
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t)>* fn = nullptr;
            size_t n = 0;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || (generation_ != seen && job_ != nullptr); });
                if (stop_) return;
                seen = generation_;
                fn = job_;
                n = total_;
                ++active_;
            }
            work(*fn, n);
            {
                std::lock_guard<std::mutex> lk(m_);
                --active_;
            }
            cv_done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_, run_mutex_;
    std::condition_variable cv_, cv_done_;
    const std::function<void(size_t)>* job_ = nullptr;
    size_t total_ = 0, done_ = 0, active_ = 0;
    std::atomic<size_t> next_{0};
    std::atomic<bool> failed_{false};
    std::exception_ptr error_;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

}  // namespace detail
}  // namespace triplet_match

#endif
