// pointcloud<Point>::content_hash (what scene<Point> keys its device copy on): the background form
// (content_hash_begin / _end, run under a search) returns what the blocking call returns, with foreground batches of the
// pool open meanwhile; one changed bit anywhere changes the hash; sizes around the slice borders. Host only.
#include <triplet_match/pointcloud.hpp>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <random>

using namespace triplet_match;
typedef PointSurfel P;

int main() {
    int failures = 0;
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    const size_t per_slice = pointcloud<P>::kHashSlice * 8 / sizeof(P); /* points per slice */
    for (size_t n : {size_t(1), size_t(7), per_slice - 1, per_slice, per_slice + 1, 5 * per_slice + 123, size_t(300000)}) {
        auto c = pointcloud<P>::empty();
        c->points.resize(n);
        for (auto& p : c->points) {
            float raw[12];
            for (float& v : raw) v = u(rng);
            std::memcpy((void*)&p, raw, sizeof raw);
        }
        const uint64_t h0 = c->content_hash();
        if (c->content_hash() != h0) ++failures; /* deterministic, whichever threads took which slices */
        pointcloud<P>::hash_job hj;
        c->content_hash_begin(hj);
        std::atomic<long> s{0};
        detail::worker_pool::instance().run(40, [&](size_t i) { s.fetch_add((long)i); }); /* a foreground batch meanwhile */
        const uint64_t h1 = c->content_hash_end(hj);
        if (h1 != h0 || s.load() != 780) ++failures;
        for (size_t at : {size_t(0), n / 2, n - 1}) { /* one bit, three places */
            uint32_t w;
            std::memcpy(&w, (const char*)&c->points[at] + 4 * (at % 12), 4);
            w ^= 1u;
            std::memcpy((char*)&c->points[at] + 4 * (at % 12), &w, 4);
            if (c->content_hash() == h0) ++failures;
            w ^= 1u;
            std::memcpy((char*)&c->points[at] + 4 * (at % 12), &w, 4);
        }
        if (c->content_hash() != h0) ++failures;
    }
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "PASSED", failures);
    return failures ? 1 : 0;
}
