// the drop-in headers' worker pool (include/triplet_match/worker_pool.hpp): every index of a batch runs exactly once,
// batches from several callers do not mix, an exception inside a task does not wedge the pool (caught or escaping), nested run() calls run inline
#include <triplet_match/worker_pool.hpp>

#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

using triplet_match::detail::worker_pool;

int main() {
    int failures = 0;
    worker_pool& pool = worker_pool::instance();
    for (size_t n : {size_t(0), size_t(1), size_t(2), size_t(7), size_t(64), size_t(1000)}) {
        for (int rep = 0; rep < 200; ++rep) {
            std::vector<std::atomic<int>> hit(n);
            for (auto& h : hit) h.store(0);
            pool.run(n, [&](size_t i) { hit[i].fetch_add(1); });
            for (size_t i = 0; i < n; ++i)
                if (hit[i].load() != 1) ++failures;
        }
    }
    {   // two callers at once: each batch sums its own indices
        std::atomic<long> sums[2];
        sums[0].store(0), sums[1].store(0);
        auto caller = [&](int who) {
            for (int rep = 0; rep < 300; ++rep) {
                std::atomic<long> s{0};
                pool.run(50, [&](size_t i) { s.fetch_add((long)i + who); });
                if (s.load() != 50 * 49 / 2 + 50 * who) sums[who].fetch_add(1);
            }
        };
        std::thread other(caller, 1);
        caller(0);
        other.join();
        failures += (int)(sums[0].load() + sums[1].load());
    }
    {   // six callers at once (searches on six host threads share the pool): batches of different lengths open side by
        // side, one of the callers' tasks throwing now and then — every batch sees each of its indices once and only
        // its own exception
        std::atomic<int> bad{0};
        auto caller = [&](int who) {
            for (int rep = 0; rep < 200; ++rep) {
                const size_t n = (size_t)(3 + 17 * who + rep % 5);
                std::vector<std::atomic<int>> hit(n);
                for (auto& h : hit) h.store(0);
                const bool throws = who == 2 && rep % 7 == 0;
                bool caught = false;
                try {
                    pool.run(n, [&](size_t i) {
                        hit[i].fetch_add(1);
                        if (throws && i == 1) throw std::runtime_error("mine");
                    });
                } catch (const std::runtime_error&) {
                    caught = true;
                }
                if (caught != throws) bad.fetch_add(1);
                for (size_t i = 0; i < n; ++i)
                    if (hit[i].load() > 1 || (!throws && hit[i].load() != 1)) bad.fetch_add(1);
            }
        };
        std::vector<std::thread> others;
        for (int who = 1; who < 6; ++who) others.emplace_back(caller, who);
        caller(0);
        for (auto& t : others) t.join();
        failures += bad.load();
    }
    {   // a task that throws is the caller's business (find_parallel catches inside the task); the pool must go on
        std::atomic<int> ran{0};
        pool.run(8, [&](size_t i) {
            try {
                if (i == 3) throw std::runtime_error("x");
                ran.fetch_add(1);
            } catch (...) {
            }
        });
        if (ran.load() != 7) ++failures;
        pool.run(8, [&](size_t) { ran.fetch_add(1); });
        if (ran.load() != 15) ++failures;
    }
    {   // a task that lets its exception escape: run() waits for every thread, rethrows the first one on the caller,
        // nothing dangles, and the pool takes the next batch (the reference's fut.get() rethrows likewise)
        for (int rep = 0; rep < 100; ++rep) {
            std::atomic<int> ran{0};
            bool caught = false;
            try {
                pool.run(64, [&](size_t i) {
                    if (i == (size_t)(rep % 64)) throw std::runtime_error("escaped");
                    ran.fetch_add(1);
                });
            } catch (const std::runtime_error& e) {
                caught = std::string(e.what()) == "escaped";
            }
            if (!caught || ran.load() > 63) ++failures;
            std::atomic<int> after{0};
            pool.run(64, [&](size_t) { after.fetch_add(1); });
            if (after.load() != 64) ++failures;
        }
    }
    {   // run() from inside a pooled task (content_hash() from a task, say) runs inline instead of deadlocking
        std::atomic<long> total{0};
        pool.run(16, [&](size_t i) {
            pool.run(10, [&](size_t j) { total.fetch_add((long)(i * 10 + j)); });
        });
        if (total.load() != 160 * 159 / 2) ++failures;
    }
    {   // a background job (the cloud's hash under a search): every index once, foreground batches opened meanwhile
        // complete first or side by side, finish() waits for all; with and without foreground work; a job object reused
        for (int rep = 0; rep < 200; ++rep) {
            const size_t n = (size_t)(1 + rep % 50);
            std::vector<std::atomic<int>> hit(n);
            for (auto& h : hit) h.store(0);
            worker_pool::job j;
            for (int twice = 0; twice < 2; ++twice) {
                pool.begin_background(j, n, [&](size_t i) { hit[i].fetch_add(1); });
                if (!j.pending()) ++failures;
                if (rep % 3) {
                    std::atomic<long> s{0};
                    pool.run(20, [&](size_t i) { s.fetch_add((long)i); });
                    if (s.load() != 190) ++failures;
                }
                pool.finish(j);
                if (j.pending()) ++failures;
                for (size_t i = 0; i < n; ++i)
                    if (hit[i].load() != twice + 1) ++failures;
            }
        }
        // a background task that throws: finish() rethrows, the pool goes on
        worker_pool::job j;
        bool caught = false;
        pool.begin_background(j, 32, [&](size_t i) {
            if (i == 5) throw std::runtime_error("bg");
        });
        try {
            pool.finish(j);
        } catch (const std::runtime_error& e) {
            caught = std::string(e.what()) == "bg";
        }
        if (!caught) ++failures;
        std::atomic<int> after{0};
        pool.run(64, [&](size_t) { after.fetch_add(1); });
        if (after.load() != 64) ++failures;
        // begun inside a pooled task: runs inline at finish()
        std::atomic<long> total{0};
        pool.run(8, [&](size_t i) {
            worker_pool::job in;
            pool.begin_background(in, 10, [&](size_t k) { total.fetch_add((long)(i * 10 + k)); });
            pool.finish(in);
        });
        if (total.load() != 80 * 79 / 2) ++failures;
    }
    std::printf("%zu workers; %s (%d failures)\n", pool.size(), failures ? "FAILED" : "PASSED", failures);
    return failures ? 1 : 0;
}
