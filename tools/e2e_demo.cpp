/* e2e_demo.cpp — times the drop-in scene<Point>::find_all_parallel on a fixture written by
 * tests/test_cpp_dropin.py::_fixture (model, scene, resolution, ground-truth poses).
 * usage: e2e_demo <fixture.bin> [early_drop 0|1] [max_icp_iterations] [repeats of the search on the kept scene]
 *        [new scene objects over the same cloud] [new clouds] [stream: up to this many host threads] [starting at] */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <memory>
#include <thread>
#include <unistd.h>

#include "triplet_match/scene.hpp"

using namespace triplet_match;
typedef PointSurfel P;

static pointcloud<P>::Ptr read_cloud(std::ifstream& in) {
    uint32_t n = 0;
    in.read((char*)&n, 4);
    auto c = pointcloud<P>::empty();
    c->points.resize(n);
    in.read((char*)c->points.data(), (std::streamsize)n * 48);
    return c;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const bool early_drop = argc > 2 ? atoi(argv[2]) != 0 : true;
    const uint32_t icp = argc > 3 ? (uint32_t)atoi(argv[3]) : 2u;
    std::ifstream in(argv[1], std::ios::binary);
    auto mcloud = read_cloud(in);
    auto scloud = read_cloud(in);
    float res = 0.f;
    in.read((char*)&res, 4);
    uint32_t n_truth = 0;
    in.read((char*)&n_truth, 4);
    std::vector<mat4f_t> truth(n_truth);
    for (auto& t : truth) in.read((char*)t.m, 64);
    mcloud->set_resolution(res);
    auto t0 = std::chrono::steady_clock::now();
    discretization_params dp;
    dp.distance_step_count = 20.f;
    dp.angle_step = 10.f * (float)M_PI / 180.f;
    model<P> m(mcloud, dp);
    sample_parameters sp;
    m.init(sp);
    auto t1 = std::chrono::steady_clock::now();
    scene<P> s(scloud);
    s.set_early_drop(early_drop);
    s.set_verbose(getenv("TM_SEARCH_VERBOSE") != nullptr);
    auto matches = s.find_all_parallel(m, 3.0f, 0.5f, 0.9f, sp, icp);
    auto t2 = std::chrono::steady_clock::now();
    printf("model %zu pts (tangent %u), scene %zu pts, planted %u\n", mcloud->size(), m.point_count(), scloud->size(), n_truth);
    printf("model init %.3f s; find_all_parallel(early_drop=%d, icp=%u) %.3f s -> %zu matches\n",
           std::chrono::duration<double>(t1 - t0).count(), (int)early_drop, icp,
           std::chrono::duration<double>(t2 - t1).count(), matches.size());
    {
        const auto& st = s.last_stats();
        printf("   phases [s]: radius %.3f shuffle %.3f sample %.3f generate+score %.3f extract %.3f icp %.3f "
               "arbitration %.3f mask %.3f | total %.3f (scene setup = %.3f)\n",
               st.radius, st.shuffle, st.sample, st.generate_score, st.extract, st.icp, st.arbitration, st.mask,
               st.total, std::chrono::duration<double>(t2 - t1).count() - st.total);
    }
    const int repeats = argc > 4 ? atoi(argv[4]) : 1;
    for (int r = 0; r < repeats; ++r) { /* the same search again: the prepared scene stays on the device between calls */
        if (repeats > 1) usleep(300000); /* a gap the kernel trace can be cut at (tools/trace_summary.py --segments) */
        auto t3 = std::chrono::steady_clock::now();
        auto again = s.find_all_parallel(m, 3.0f, 0.5f, 0.9f, sp, icp);
        auto t4 = std::chrono::steady_clock::now();
        printf("%s call %.4f s -> %zu matches\n", r == 0 ? "second" : "further", std::chrono::duration<double>(t4 - t3).count(),
               again.size());
        const auto& st = s.last_stats();
        printf("   phases [ms]: radius %.2f sample %.2f generate+score %.2f extract %.2f icp %.2f arbitration %.2f mask %.2f | "
               "total %.2f\n", 1e3 * st.radius, 1e3 * st.sample, 1e3 * st.generate_score, 1e3 * st.extract, 1e3 * st.icp,
               1e3 * st.arbitration, 1e3 * st.mask, 1e3 * st.total);
        printf("   sampling: %llu rounds, %llu first points sampled, %llu positions drawn, %llu neighbourhood entries handed over\n",
               (unsigned long long)st.rounds, (unsigned long long)st.firsts_sampled, (unsigned long long)st.draws,
               (unsigned long long)st.neighbours);
    }
    if (argc > 5) { /* a NEW scene object over the same cloud each time: what one frame of a stream costs */
        const int frames = atoi(argv[5]);
        for (int f = 0; f < frames; ++f) {
            auto t5 = std::chrono::steady_clock::now();
            scene<P> fresh(scloud);
            fresh.set_early_drop(early_drop);
            auto got = fresh.find_all_parallel(m, 3.0f, 0.5f, 0.9f, sp, icp);
            auto t6 = std::chrono::steady_clock::now();
            printf("new scene %d: %.4f s -> %zu matches (search itself %.2f ms)\n", f, std::chrono::duration<double>(t6 - t5).count(),
                   got.size(), 1e3 * fresh.last_stats().total);
        }
    }
    if (argc > 6) { /* a NEW cloud (one coordinate nudged: other content) and a new scene per call: a frame of a live
                     * stream, nothing cached — upload, curve order, curvature of the tangent candidates, masks, search */
        const int frames = atoi(argv[6]);
        for (int f = 0; f < frames; ++f) {
            auto fcloud = pointcloud<P>::empty();
            fcloud->points = scloud->points;
            fcloud->points[(size_t)f * 7919u % fcloud->points.size()].x += 1e-4f;
            auto t5 = std::chrono::steady_clock::now();
            scene<P> fresh(fcloud);
            fresh.set_early_drop(early_drop);
            fresh.set_verbose(getenv("TM_FRAME_VERBOSE") != nullptr);
            auto got = fresh.find_all_parallel(m, 3.0f, 0.5f, 0.9f, sp, icp);
            auto t6 = std::chrono::steady_clock::now();
            printf("new cloud %d: %.4f s -> %zu matches (search itself %.2f ms)\n", f, std::chrono::duration<double>(t6 - t5).count(),
                   got.size(), 1e3 * fresh.last_stats().total);
        }
    }
    if (argc > 7) { /* a stream of new clouds worked off by T host threads, each with a context and a model of its own
                     * (make_context): the threads' kernels run side by side on the GPU and nothing is shared but the
                     * headers' worker pool — frames per second for T = 1 .. argv[7]. (Threads on the one default context
                     * serialise on its lock: 54 -> 61 frames/s with two.) */
        const int max_threads = atoi(argv[7]);
        const int frames = 32;
        std::vector<pointcloud<P>::Ptr> clouds;
        for (int f = 0; f < frames; ++f) {
            auto c = pointcloud<P>::empty();
            c->points = scloud->points;
            c->points[(size_t)(f + 100) * 7919u % c->points.size()].y += 1e-4f;
            clouds.push_back(c);
        }
        std::vector<context_t> ctxs;
        std::vector<std::unique_ptr<model<P>>> models;
        for (int t = 0; t < max_threads; ++t) {
            ctxs.push_back(make_context(0));
            models.emplace_back(new model<P>(mcloud, dp, ctxs.back()));
            models.back()->init(sp);
        }
        const int first_threads = argc > 8 ? std::max(1, atoi(argv[8])) : 1;
        for (int T = first_threads; T <= max_threads; ++T) {
            usleep(300000); /* a gap the kernel trace can be cut at (tools/trace_summary.py --segments) */
            std::atomic<int> next{0}, found{0};
            auto t7 = std::chrono::steady_clock::now();
            std::vector<std::thread> pool;
            for (int t = 0; t < T; ++t)
                pool.emplace_back([&, t]() {
                    for (;;) {
                        const int f = next.fetch_add(1);
                        if (f >= frames) break;
                        scene<P> fresh(clouds[f], ctxs[t]);
                        fresh.set_early_drop(early_drop);
                        found.fetch_add((int)fresh.find_all_parallel(*models[t], 3.0f, 0.5f, 0.9f, sp, icp).size());
                    }
                });
            for (auto& th : pool) th.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t7).count();
            printf("stream: %d frames on %d thread%s, a context each: %.1f frames/s (%.2f ms per frame), %d matches\n", frames, T,
                   T > 1 ? "s" : "", frames / dt, 1e3 * dt / frames, found.load());
            for (auto& c : clouds) c->touch(); /* the next round starts from clouds without cached curvature */
        }
    }
    for (const auto& mt : matches) {
        double best = 1e30;
        for (const auto& t : truth) {
            mat4f_t m2s = t.inverse();
            double err = 0;
            for (size_t k = 0; k < mcloud->size(); k += 97) {
                vec4f_t a = mt.transform * homogeneous(mcloud->points[k].getVector3fMap());
                vec4f_t b = m2s * homogeneous(mcloud->points[k].getVector3fMap());
                err = std::max(err, (double)(a.head3() - b.head3()).norm());
            }
            best = std::min(best, err);
        }
        printf("   match: %zu correspondences, score %.4f, max alignment error %.5f (= %.2f x resolution)\n",
               mt.scene_corrs.size(), mt.signed_score, best, best / res);
    }
    return 0;
}
