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
 * out. run(n, fn) calls fn(0 .. n-1), the caller taking part, and returns when all are done. Batches of different
 * callers are open side by side (searches on several host threads, each with its context, share these workers: with
 * one batch at a time a search waited for another search's hash before it could sample its own chunks); a worker
 * takes indices of the oldest open batch. The threads are created on first use and joined when the process exits.
 * A task that throws (on the caller's thread or on a worker) does not tear anything down: the first exception of its
 * batch is kept, the indices not yet started are passed over, run() waits until every thread has left the batch and
 * rethrows on the caller — as the reference's fut.get() would (:168-171). run() from inside a task (a pooled task
 * that hashes a cloud, say) runs inline on that thread instead of waiting for the pool it is part of.
 * begin_background(job, n, fn) / finish(job): the same, split in two — the workers take the job's indices whenever no
 * run() batch is open (those go first), the caller carries on and joins in at finish(), which returns when all are done
 * (the cloud's content hash of a search runs under the search this way). */
class worker_pool {
    struct batch {
        const std::function<void(size_t)>* fn = nullptr;
        size_t n = 0;
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        size_t done = 0, active = 0; /* under m_: indices finished; workers (not the caller) inside the batch */
        std::exception_ptr error;    /* under m_ */
        bool background = false;
    };

public:
    /* a background job: lives where the caller keeps it, from begin_background() until finish() has returned */
    class job {
        friend class worker_pool;
        batch b;
        std::function<void(size_t)> fn;
        bool queued = false, begun = false;

    public:
        job() = default;
        job(const job&) = delete;
        job& operator=(const job&) = delete;
        bool pending() const { return begun; }
    };
    void begin_background(job& j, size_t n, std::function<void(size_t)> fn) {
        j.fn = std::move(fn);
        j.b.fn = &j.fn;
        j.b.n = n;
        j.b.background = true;
        j.b.next.store(0);
        j.b.failed.store(false);
        j.b.done = 0, j.b.active = 0;
        j.b.error = nullptr;
        j.begun = true;
        j.queued = false;
        if (n == 0 || threads_.empty() || in_task()) return; /* finish() runs it inline */
        {
            std::lock_guard<std::mutex> lk(m_);
            open_.push_back(&j.b);
        }
        j.queued = true;
        cv_.notify_all();
    }
    void finish(job& j) {
        if (!j.begun) return;
        j.begun = false;
        if (!j.queued) {
            for (size_t i = 0; i < j.b.n; ++i) j.fn(i);
            return;
        }
        work(j.b);
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> lk(m_);
            close(j.b);
            cv_done_.wait(lk, [&] { return j.b.done == j.b.n && j.b.active == 0; });
            err = j.b.error;
        }
        j.queued = false;
        if (err) std::rethrow_exception(err);
    }

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
        batch b;
        b.fn = &fn;
        b.n = n;
        {
            std::lock_guard<std::mutex> lk(m_);
            auto at = open_.begin(); /* after the open run() batches, before the background jobs */
            while (at != open_.end() && !(*at)->background) ++at;
            open_.insert(at, &b);
        }
        cv_.notify_all();
        work(b);
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> lk(m_);
            close(b); /* every index has been handed out */
            cv_done_.wait(lk, [&] { return b.done == b.n && b.active == 0; });
            err = b.error;
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
    /* m_ held: the batch takes no more workers */
    void close(batch& b) {
        auto it = std::find(open_.begin(), open_.end(), &b);
        if (it != open_.end()) open_.erase(it);
    }
    void work(batch& b, size_t limit = ~(size_t)0) {
        size_t finished = 0;
        std::exception_ptr err;
        bool& nested = in_task();
        nested = true;
        for (;;) {
            const size_t i = b.next.fetch_add(1);
            if (i >= b.n) break;
            if (!b.failed.load(std::memory_order_relaxed)) {
                try {
                    (*b.fn)(i);
                } catch (...) {
                    if (!err) err = std::current_exception();
                    b.failed.store(true);
                }
            }
            ++finished; /* an index passed over after a failure still counts: run() waits for done == n */
            if (finished >= limit) break;
        }
        nested = false;
        if (finished || err) {
            std::lock_guard<std::mutex> lk(m_);
            b.done += finished;
            if (err && !b.error) b.error = err;
        }
    }
    void loop() {
        for (;;) {
            batch* b = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !open_.empty(); });
                if (stop_) return;
                b = open_.front();
                ++b->active; /* its caller does not return before this is back at zero */
            }
            /* a background job one index at a time: a run() batch that opens meanwhile is served first */
            work(*b, b->background ? 1 : ~(size_t)0);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (b->next.load() >= b->n) close(*b); /* no index left */
                --b->active;
            }
            cv_done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    std::vector<batch*> open_; /* batches that still hand out indices, oldest first */
    bool stop_ = false;
};

}  // namespace detail
}  // namespace triplet_match

#endif
