/*
 * triplet_match/scene.hpp — scene<Point> of the drop-in API over the C ABI.
 * Mirrors include/scene:10-59 / include/impl/scene.hpp: same constructor, find_all_parallel(),
 * instantiate(), cloud() and match_t. The search orchestration is search.hpp (host logic); descriptor
 * match, hypothesis generation, inlier scoring and correspondence extraction run as HIP kernels.
 */
#ifndef TRIPLET_MATCH_SCENE_HPP_
#define TRIPLET_MATCH_SCENE_HPP_

#include <algorithm>
#include <chrono>
#include <cstdlib>

#include "model.hpp"
#include "projected_search.hpp"
#include "search.hpp"

namespace triplet_match {

namespace detail {

/* Host-side cross-check switches of the search driver (three calls per batch instead of tm_search_batch, one call
 * per chunk, the general subset call, the ICP loop on the host): same results either way. Read from the
 * environment ONCE per process, on first use — never inside the search loops. */
struct search_switches {
    bool search_unfused, search_per_chunk, subset_general, icp_host_loop, icp_per_candidate, dense_neighbourhoods,
        host_curvature_tree, sequential_sampling, hash_first, host_arbitration;
};
inline const search_switches& env_once() {
    static const search_switches sw = {std::getenv("TM_SEARCH_UNFUSED") != nullptr,
                                       std::getenv("TM_SEARCH_PER_CHUNK") != nullptr,
                                       std::getenv("TM_SUBSET_GENERAL") != nullptr,
                                       std::getenv("TM_ICP_HOST_LOOP") != nullptr,
                                       std::getenv("TM_ICP_PER_CANDIDATE") != nullptr,
                                       std::getenv("TM_DENSE_NEIGHBOURHOODS") != nullptr,
                                       std::getenv("TM_HOST_CURVATURE_TREE") != nullptr,
                                       std::getenv("TM_SEQUENTIAL_SAMPLING") != nullptr,
                                       std::getenv("TM_HASH_FIRST") != nullptr,
                                       std::getenv("TM_HOST_ARBITRATION") != nullptr};
    return sw;
}

/* search.hpp backend over libtm_hip.so */
template <typename Point>
struct hip_backend {
    tm_scene* s;
    tm_model* m;
    const pointcloud<Point>* scloud;
    const pointcloud<Point>* mcloud;
    uint32_t n_t;
    float diam;
    int traits_kind = TM_TRAITS_IDENTITY; /* the model's metric (tm_model_set_metric) */
    search::params_t p;
    std::vector<uint32_t> sc_scratch, mc_scratch; /* extract output buffers */

    uint32_t scene_size() const { return (uint32_t)scloud->size(); }
    uint32_t model_size() const { return (uint32_t)mcloud->size(); }
    uint32_t model_point_count() const { return n_t; }
    float model_diameter() const { return diam; }
    const float* scene_surfel(uint32_t i) const { return scloud->points[i].raw(); }
    const float* model_surfel(uint32_t i) const { return mcloud->points[i].raw(); }
    /* raw_rows: the scene on the device holds exactly scloud's surfels (the identity search): the filters of
     * include/impl/scene.hpp:293-302 then run on the caller's rows through the pure tm_pair_valid_raw — no handle, no
     * lock, callable from several sampling threads (the loop has done the mask tests of :290 before it asks). A
     * projected scene's rows live in the library: tm_pair_is_valid_sample. */
    bool raw_rows = false;
    float bmin[4] = {0.f, 0.f, 0.f, 0.f}, bmax[4] = {0.f, 0.f, 0.f, 0.f};
    bool pair_is_valid_sample(uint32_t i, uint32_t j) const {
        if (raw_rows)
            return i != j && tm_pair_valid_raw(scloud->points[i].raw(), scloud->points[j].raw(), diam, p.min_diameter_factor,
                                               p.max_diameter_factor, bmin, bmax) != 0;
        const int rc = tm_pair_is_valid_sample(s, m, i, j, p.min_diameter_factor, p.max_diameter_factor);
        if (rc < 0) check(rc, "tm_pair_is_valid_sample"); /* throws: the scene's rows could not be read */
        return rc != 0;
    }
    /* radius_search_inclusive(radius, i), include/impl/scene.hpp:273: the ascending index lists are built on
     * the device for up to 16 of the announced first points per call (announce_firsts) and handed out when
     * the loop reaches each point */
    std::vector<uint32_t> upcoming;   /* first points of the current find_in_subset call */
    std::vector<int32_t> list_cache;  /* lists of upcoming[cache_lo .. cache_hi), cache_stride apart */
    std::vector<uint32_t> count_cache;
    size_t cache_lo = 0, cache_hi = 0, cache_stride = 0;
    float cache_radius = -1.f;
    bool round_announced = false; /* announce_round(): the first points of ALL chunks of a find_parallel round */
    void announce_firsts(const std::vector<uint32_t>& firsts) {
        if (round_announced && !firsts.empty() &&
            std::find(upcoming.begin(), upcoming.end(), firsts[0]) != upcoming.end())
            return; /* a chunk of the announced round: its points are already in the queue */
        round_announced = false;
        upcoming = firsts;
        cache_lo = cache_hi = 0;
        t_lo = t_hi = 0;
    }
    /* every chunk's first points up front, in visit order: the neighbourhoods are then fetched for up to 64 first
     * points per call across chunk borders (an empty list ends the round) */
    void announce_round(const std::vector<uint32_t>& firsts) {
        upcoming = firsts;
        cache_lo = cache_hi = 0;
        t_lo = t_hi = 0;
        round_announced = !firsts.empty();
    }
    /* The sampling loop's view of nn = radius_search_inclusive(radius, i) (include/impl/scene.hpp:273-302): its
     * size and (rank, index) of its members with tangent_mask set (tm_scene_radius_tangent_lists) — every other
     * member fails the first filter of :290. Fetched for i and up to 63 announced first points after it. */
    std::vector<uint32_t> t_counts, t_offsets, t_pairs; /* per cached first point: |nn|, CSR of (rank, index), ascending */
    size_t t_lo = 0, t_hi = 0;
    float t_radius = -1.f;
    bool supports_tangent_view() const { return !env_once().dense_neighbourhoods; }
    void fetch_tangent_lists(uint32_t i, float radius) {
        size_t pos = 0;
        while (pos < upcoming.size() && upcoming[pos] != i) ++pos;
        std::vector<uint32_t> batch;
        if (pos < upcoming.size()) {
            for (size_t q = pos; q < upcoming.size() && batch.size() < 64; ++q) batch.push_back(upcoming[q]);
            t_lo = pos;
        } else {
            batch.push_back(i);
            upcoming.assign(1, i);
            round_announced = false;
            t_lo = 0;
        }
        t_hi = t_lo + batch.size();
        t_radius = radius;
        t_counts.assign(batch.size(), 0u);
        std::vector<uint32_t> tc(batch.size(), 0u);
        uint32_t n_pairs_total = 0;
        if (t_pairs.size() < 2 * 8192) t_pairs.resize(2 * 8192);
        for (int attempt = 0; attempt < 2; ++attempt) {
            const int rc = tm_scene_radius_tangent_lists(s, batch.data(), (uint32_t)batch.size(), radius, t_counts.data(),
                                                         tc.data(), t_pairs.data(), (uint32_t)(t_pairs.size() / 2),
                                                         &n_pairs_total);
            if (rc == TM_ERR_CAPACITY && attempt == 0) {
                t_pairs.resize(2 * ((size_t)n_pairs_total + n_pairs_total / 4 + 16));
                continue;
            }
            check(rc, "tm_scene_radius_tangent_lists");
            break;
        }
        /* the pairs arrive slot after slot, each slot in ascending index (= ascending rank) order */
        t_offsets.assign(batch.size() + 1, 0u);
        for (size_t k = 0; k < batch.size(); ++k) t_offsets[k + 1] = t_offsets[k] + tc[k];
    }
    /* search::find_parallel: fetches the views of the announced round; true = every first point of the round is in the
     * window, so tangent_view() of any of them only reads, and the validity test runs without the library
     * (raw_rows): the chunks may be sampled on concurrent threads */
    bool round_views_ready(float radius) {
        if (!raw_rows || !round_announced || upcoming.empty() || !supports_tangent_view() || env_once().sequential_sampling)
            return false;
        if (!(t_radius == radius && t_lo == 0 && t_hi == upcoming.size())) fetch_tangent_lists(upcoming[0], radius);
        return t_radius == radius && t_lo == 0 && t_hi == upcoming.size();
    }
    void tangent_view(uint32_t i, float radius, size_t& n_nn, const uint32_t*& rank_index_pairs, size_t& n_pairs) {
        size_t k = t_lo;
        while (k < t_hi && upcoming[k] != i) ++k;
        if (k >= t_hi || t_radius != radius) {
            fetch_tangent_lists(i, radius);
            k = t_lo;
        }
        n_nn = t_counts[k - t_lo];
        rank_index_pairs = t_pairs.data() + 2 * (size_t)t_offsets[k - t_lo];
        n_pairs = t_offsets[k - t_lo + 1] - t_offsets[k - t_lo];
    }
    void fetch_lists(uint32_t i, float radius) { /* i and the announced firsts that follow it */
        {
            size_t pos = 0;
            while (pos < upcoming.size() && upcoming[pos] != i) ++pos;
            std::vector<uint32_t> batch;
            if (pos < upcoming.size()) {
                for (size_t q = pos; q < upcoming.size() && batch.size() < 16; ++q) batch.push_back(upcoming[q]);
                cache_lo = pos;
            } else { /* a point that was not announced */
                batch.push_back(i);
                upcoming.assign(1, i);
                cache_lo = 0;
            }
            cache_hi = cache_lo + batch.size();
            cache_radius = radius;
            count_cache.assign(batch.size(), 0u);
            /* one call with the stride of the last batch (+25 %); a larger neighbourhood reports
             * TM_ERR_CAPACITY with the counts filled in, and the call is repeated with room for it */
            if (cache_stride < 1024) cache_stride = 1024;
            for (int attempt = 0; attempt < 2; ++attempt) {
                if (list_cache.size() < batch.size() * cache_stride) list_cache.resize(batch.size() * cache_stride);
                const int rc = tm_scene_radius_lists(s, batch.data(), (uint32_t)batch.size(), radius, list_cache.data(),
                                                     (uint32_t)cache_stride, count_cache.data());
                if (rc == TM_ERR_CAPACITY && attempt == 0) {
                    size_t mx = 1;
                    for (uint32_t cnt : count_cache) mx = std::max<size_t>(mx, cnt);
                    cache_stride = mx + mx / 4;
                    continue;
                }
                check(rc, "tm_scene_radius_lists");
                break;
            }
        }
    }
    void radius_subset(uint32_t i, float radius, std::vector<int32_t>& out) {
        size_t k = cache_lo;
        while (k < cache_hi && upcoming[k] != i) ++k;
        if (k >= cache_hi || cache_radius != radius) { /* not cached */
            fetch_lists(i, radius);
            k = cache_lo;
        }
        const int32_t* src = list_cache.data() + (k - cache_lo) * cache_stride;
        out.assign(src, src + count_cache[k - cache_lo]);
    }
    /* the same list in place: valid until the next radius_subset / radius_view call */
    void radius_view(uint32_t i, float radius, const int32_t*& ptr, size_t& n) {
        size_t k = cache_lo;
        while (k < cache_hi && upcoming[k] != i) ++k;
        if (k >= cache_hi || cache_radius != radius) {
            fetch_lists(i, radius);
            k = cache_lo;
        }
        ptr = list_cache.data() + (k - cache_lo) * cache_stride;
        n = count_cache[k - cache_lo];
    }
    void generate(const std::vector<uint32_t>& ij, std::vector<search::pose_t>& poses) {
        const uint32_t n_pairs = (uint32_t)(ij.size() / 2);
        const uint32_t cap = n_pairs * (uint32_t)search::query_limit;
        poses.resize(cap);
        std::vector<uint32_t> prov(4 * (size_t)cap);
        uint32_t count = 0, total = 0;
        check(tm_generate_hypotheses(s, m, ij.data(), n_pairs, p.min_diameter_factor, p.max_diameter_factor,
                                     p.force_up ? 1 : 0, search::query_limit, &poses[0].m[0], prov.data(), cap,
                                     &count, &total),
              "tm_generate_hypotheses");
        poses.resize(count);
    }
    bool supports_batched() const { return true; }
    void generate_with_pairs(const std::vector<uint32_t>& ij, std::vector<search::pose_t>& poses,
                             std::vector<uint32_t>& pair_of_hyp) {
        const uint32_t n_pairs = (uint32_t)(ij.size() / 2);
        const uint32_t cap = n_pairs * (uint32_t)search::query_limit;
        poses.resize(cap);
        std::vector<uint32_t> prov(4 * (size_t)cap);
        uint32_t count = 0, total = 0;
        check(tm_generate_hypotheses(s, m, ij.data(), n_pairs, p.min_diameter_factor, p.max_diameter_factor,
                                     p.force_up ? 1 : 0, search::query_limit, &poses[0].m[0], prov.data(), cap,
                                     &count, &total),
              "tm_generate_hypotheses");
        poses.resize(count);
        pair_of_hyp.resize(count);
        for (uint32_t h = 0; h < count; ++h) pair_of_hyp[h] = prov[4 * (size_t)h];
    }
    /* one batch of find_in_subset on the device (tm_search_batch); TM_SEARCH_UNFUSED=1: the three calls above
     * and below with the scan on the host (same results) */
    bool supports_fused_batch() const { return !env_once().search_unfused; }
    bool search_batch(const std::vector<uint32_t>& ij, float radius, bool early_drop, float sufficient,
                      search::pose_t& pose) {
        tm_batch_result r;
        check(tm_search_batch(s, m, ij.data(), (uint32_t)(ij.size() / 2), p.min_diameter_factor, p.max_diameter_factor,
                              p.force_up ? 1 : 0, search::query_limit, radius, p.dist_thres, p.accept_prob,
                              early_drop ? 1 : 0, sufficient, &r),
              "tm_search_batch");
        if (r.index < 0) return false;
        for (int k = 0; k < 16; ++k) pose.m[k] = r.pose[k];
        return true;
    }
    /* the batches of a whole find_parallel round in one call (tm_search_batches); TM_SEARCH_PER_CHUNK=1: one call
     * per batch as the chunks come */
    bool supports_deferred_batches() const { return supports_fused_batch() && !env_once().search_per_chunk; }
    void search_batches(const std::vector<std::vector<uint32_t>>& batches, float radius, bool early_drop,
                        float sufficient, std::vector<char>& found, std::vector<search::pose_t>& poses) {
        std::vector<uint32_t> ij, offsets(1, 0u);
        for (const auto& b : batches) {
            ij.insert(ij.end(), b.begin(), b.end());
            offsets.push_back((uint32_t)(ij.size() / 2));
        }
        found.assign(batches.size(), 0);
        poses.resize(batches.size());
        if (batches.empty()) return;
        std::vector<tm_batch_result> r(batches.size());
        check(tm_search_batches(s, m, ij.data(), (uint32_t)(ij.size() / 2), offsets.data(), (uint32_t)batches.size(),
                                p.min_diameter_factor, p.max_diameter_factor, p.force_up ? 1 : 0, search::query_limit,
                                radius, p.dist_thres, p.accept_prob, early_drop ? 1 : 0, sufficient, r.data()),
              "tm_search_batches");
        for (size_t b = 0; b < batches.size(); ++b)
            if (r[b].index >= 0) {
                found[b] = 1;
                for (int k = 0; k < 16; ++k) poses[b].m[k] = r[b].pose[k];
            }
    }
    void score_ball(const std::vector<search::pose_t>& poses, const std::vector<uint32_t>& centers, float radius,
                    bool early_drop, std::vector<uint32_t>& counts, std::vector<double>& scores) {
        counts.assign(poses.size(), 0u);
        scores.assign(poses.size(), 0.0);
        if (early_drop) {
            std::vector<int32_t> dropped(poses.size());
            std::vector<uint32_t> saved(poses.size());
            check(tm_score_hypotheses_early_drop_ball(s, m, &poses[0].m[0], (uint32_t)poses.size(), centers.data(),
                                                      radius, p.dist_thres, p.accept_prob, counts.data(),
                                                      scores.data(), dropped.data(), saved.data()),
                  "tm_score_hypotheses_early_drop_ball");
        } else {
            check(tm_score_hypotheses_ball(s, m, &poses[0].m[0], (uint32_t)poses.size(), centers.data(), radius,
                                           p.dist_thres, counts.data(), scores.data()),
                  "tm_score_hypotheses_ball");
        }
    }
    void score(const std::vector<search::pose_t>& poses, const std::vector<int32_t>& subset, bool early_drop,
               std::vector<uint32_t>& counts, std::vector<double>& scores) {
        const uint32_t n = (uint32_t)poses.size();
        counts.assign(n, 0u);
        scores.assign(n, 0.0);
        if (early_drop) {
            std::vector<int32_t> dropped(n);
            std::vector<uint32_t> saved(n);
            check(tm_score_hypotheses_early_drop(s, m, &poses[0].m[0], n, subset.data(), (uint32_t)subset.size(),
                                                 p.dist_thres, p.accept_prob, counts.data(), scores.data(),
                                                 dropped.data(), saved.data()),
                  "tm_score_hypotheses_early_drop");
        } else {
            /* a short list of points and few poses: one thread per (pose, point) instead of a walk over the scene
             * per pose (a list with a repeated point is refused and takes the general call) */
            int rc = TM_ERR_INVALID;
            if (n <= 65535u && !subset.empty() && (uint64_t)n * subset.size() <= (1ull << 28) &&
                !env_once().subset_general)
                rc = tm_score_hypotheses_on_points(s, m, &poses[0].m[0], n, subset.data(), (uint32_t)subset.size(),
                                                   p.dist_thres, counts.data(), scores.data());
            if (rc != TM_OK)
                check(tm_score_hypotheses(s, m, traits_kind, &poses[0].m[0], n, subset.data(),
                                          (uint32_t)subset.size(), p.dist_thres, TM_SCORE_AUTO, counts.data(),
                                          scores.data()),
                      "tm_score_hypotheses");
        }
    }
    void extract(const search::pose_t& T, float dist_thres, search::candidate_t& out) {
        const uint32_t cap = (uint32_t)scloud->size();
        out.transform = T;
        sc_scratch.resize(cap);
        mc_scratch.resize(cap);
        uint32_t count = 0;
        check(tm_extract_correspondences(s, m, T.m, dist_thres, sc_scratch.data(), mc_scratch.data(), cap, &count,
                                         &out.signed_score),
              "tm_extract_correspondences");
        out.scene_corrs.assign(sc_scratch.begin(), sc_scratch.begin() + count);
        out.model_corrs.assign(mc_scratch.begin(), mc_scratch.begin() + count);
    }
    /* finish_find of several poses in one call (tm_extract_correspondences_batch): one read-back of the counts, one
     * of the lists */
    void extract_batch(const std::vector<search::pose_t>& Ts, float dist_thres, std::vector<search::candidate_t>& out) {
        const uint32_t n = (uint32_t)Ts.size();
        out.resize(n);
        if (n == 0) return;
        std::vector<uint32_t> counts(n);
        std::vector<double> scores(n);
        /* rows per pose: what the last call needed (kept across searches), or at first a little more than the model's
         * size — a good pose matches about one scene point per model point; a short guess costs a second pass */
        size_t stride = batch_stride ? batch_stride : std::max<size_t>(1024, mcloud ? mcloud->size() + mcloud->size() / 4 : 0);
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (sc_scratch.size() < stride * n) sc_scratch.resize(stride * n);
            if (mc_scratch.size() < stride * n) mc_scratch.resize(stride * n);
            const int rc = tm_extract_correspondences_batch(s, m, &Ts[0].m[0], n, dist_thres, sc_scratch.data(),
                                                            mc_scratch.data(), (uint32_t)stride, counts.data(), scores.data());
            if (rc == TM_ERR_CAPACITY && attempt == 0) {
                size_t mx = 1;
                for (uint32_t cnt : counts) mx = std::max<size_t>(mx, cnt);
                stride = batch_stride = mx + mx / 8;
                continue;
            }
            check(rc, "tm_extract_correspondences_batch");
            break;
        }
        /* the candidates' lists copied side by side (0.4 MB each way per candidate) */
        ::triplet_match::detail::worker_pool::instance().run(n, [&](size_t k) {
            out[k].transform = Ts[k];
            out[k].signed_score = scores[k];
            out[k].scene_corrs.assign(sc_scratch.begin() + k * stride, sc_scratch.begin() + k * stride + counts[k]);
            out[k].model_corrs.assign(mc_scratch.begin() + k * stride, mc_scratch.begin() + k * stride + counts[k]);
        });
    }
    size_t batch_stride = 0;
    /* finish_find of a round's candidates AND their arbitration (include/impl/scene.hpp:193-237) where the lists are
     * (tm_extract_arbitrate_batch): only the accepted candidates' lists come back, the greedy pass over half a million
     * list entries does not run on this side, and the accepted points reach the device's mask_ without going up again
     * (add_extracted_to_mask). Set by scene<Point>'s identity search, whose tangent mask IS the device scene's;
     * TM_HOST_ARBITRATION=1 keeps search::arbitrate. false = not decided here: the caller extracts and arbitrates. */
    bool device_arbitration = false;
    bool extract_arbitrate(const std::vector<search::pose_t>& Ts, float dist_thres, uint32_t min_points, uint32_t min_points_all,
                           std::vector<search::candidate_t>& accepted, bool verbose) {
        accepted.clear();
        if (!device_arbitration || env_once().host_arbitration) return false;
        const uint32_t n = (uint32_t)Ts.size();
        if (n == 0) return true;
        std::vector<uint32_t> counts(n), kept(n), tan(n);
        std::vector<double> scores(n);
        std::vector<int32_t> verdict(n, -1);
        size_t stride = batch_stride ? batch_stride : std::max<size_t>(1024, mcloud ? mcloud->size() + mcloud->size() / 4 : 0);
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (sc_scratch.size() < stride * n) sc_scratch.resize(stride * n);
            if (mc_scratch.size() < stride * n) mc_scratch.resize(stride * n);
            const int rc = tm_extract_arbitrate_batch(s, m, &Ts[0].m[0], n, dist_thres, min_points, min_points_all, sc_scratch.data(),
                                                      mc_scratch.data(), (uint32_t)stride, counts.data(), scores.data(),
                                                      verdict.data(), kept.data(), tan.data());
            if (rc == TM_ERR_CAPACITY && attempt == 0) {
                size_t mx = 1;
                for (uint32_t cnt : counts) mx = std::max<size_t>(mx, cnt);
                stride = batch_stride = mx + mx / 8;
                continue;
            }
            check(rc, "tm_extract_arbitrate_batch");
            break;
        }
        for (int32_t v : verdict)
            if (v < 0) return false; /* an accepted candidate would have to be trimmed: the host's pass does that */
        std::vector<uint32_t> order(n);
        for (uint32_t k = 0; k < n; ++k) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return counts[x] > counts[y]; }); /* :193-195 */
        std::vector<uint32_t> take;
        for (uint32_t k : order) {
            if (verbose)
                std::fprintf(stderr, "[search] candidate: %u corrs (%u after overlap removal), tangent inliers %u (need %u), all %u (need %u)\n",
                             counts[k], kept[k], tan[k], min_points, kept[k], min_points_all);
            if (verdict[k] == 1) take.push_back(k);
        }
        accepted.resize(take.size());
        ::triplet_match::detail::worker_pool::instance().run(take.size(), [&](size_t a) {
            const uint32_t k = take[a];
            accepted[a].transform = Ts[k];
            accepted[a].signed_score = scores[k];
            accepted[a].device_row = (int32_t)k;
            accepted[a].scene_corrs.assign(sc_scratch.begin() + k * stride, sc_scratch.begin() + k * stride + counts[k]);
            accepted[a].model_corrs.assign(mc_scratch.begin() + k * stride, mc_scratch.begin() + k * stride + counts[k]);
        });
        return true;
    }
    void add_extracted_to_mask(const std::vector<uint32_t>& rows) {
        check(tm_scene_mask_add_extracted(s, rows.data(), (uint32_t)rows.size()), "tm_scene_mask_add_extracted");
    }
    /* the whole ICP loop on the device (tm_icp_refine); TM_ICP_HOST_LOOP=1: one call per pass, fits on the host */
    bool supports_icp_refine() const { return !env_once().icp_host_loop; }
    search::pose_t icp_refine(const search::pose_t& T, float dist_thres, uint32_t max_iterations) {
        search::pose_t out;
        uint32_t n = 0, it = 0;
        check(tm_icp_refine(s, m, T.m, dist_thres, max_iterations, out.m, &n, &it), "tm_icp_refine");
        return out;
    }
    /* the candidates of one find_parallel round refined side by side (tm_icp_refine_batch); TM_ICP_PER_CANDIDATE=1:
     * one tm_icp_refine call per candidate, as the reference's loop goes (include/impl/scene.hpp:169-171) */
    bool supports_icp_refine_batch() const { return supports_icp_refine() && !env_once().icp_per_candidate; }
    void icp_refine_batch(const std::vector<search::pose_t>& starts, float dist_thres, uint32_t max_iterations,
                          std::vector<search::pose_t>& out) {
        out.resize(starts.size());
        if (starts.empty()) return;
        std::vector<uint32_t> counts(starts.size()), iters(starts.size());
        check(tm_icp_refine_batch(s, m, &starts[0].m[0], (uint32_t)starts.size(), dist_thres, max_iterations,
                                  &out[0].m[0], counts.data(), iters.data()),
              "tm_icp_refine_batch");
    }
    search::detail::icp_sums_t icp_sums(const search::pose_t& T, float dist_thres) {
        search::detail::icp_sums_t r;
        check(tm_icp_accumulate(s, m, T.m, dist_thres, &r.n, r.v), "tm_icp_accumulate");
        return r;
    }
    void set_mask(const std::vector<int32_t>& mask) { check(tm_scene_set_mask(s, mask.data()), "tm_scene_set_mask"); }
    /* the two changes find_all makes to mask_ (:41 all clear, :87-90 an accepted match's points) without moving the
     * whole mask: a device memset (nothing at all when it is clear already), the new indices only */
    void clear_mask() { check(tm_scene_set_mask(s, nullptr), "tm_scene_set_mask"); }
    void add_to_mask(const std::vector<uint32_t>& idx) {
        check(tm_scene_mask_add(s, idx.data(), (uint32_t)idx.size()), "tm_scene_mask_add");
    }
};

/* search.hpp Exchange over the C ABI's RCCL communicator */
struct hip_exchange {
    tm_comm* c;
    int rank() const { return tm_comm_rank(c); }
    int world() const { return tm_comm_world(c); }
    void allgather(const search::chunk_record* local, uint32_t n, search::chunk_record* all) const {
        static_assert(sizeof(search::chunk_record) == sizeof(tm_best_record), "record layouts must agree");
        check(tm_comm_allgather_records(c, reinterpret_cast<const tm_best_record*>(local), n,
                                        reinterpret_cast<tm_best_record*>(all)),
              "tm_comm_allgather_records");
    }
};

}  // namespace detail

template <typename Point>
class scene {
public:
    typedef std::unique_ptr<scene<Point>> uptr_t;
    typedef std::shared_ptr<scene<Point>> sptr_t;
    typedef std::weak_ptr<scene<Point>> wptr_t;
    typedef std::weak_ptr<const scene<Point>> const_wptr_t;
    typedef model<Point> model_t;
    typedef pointcloud<Point> cloud_t;

    typedef struct match_ { /* include/scene:23-28 */
        mat4f_t transform;
        subset_t scene_corrs;
        subset_t model_corrs;
        double signed_score;
    } match_t;

    struct statistics {
        double rejection_rate;
    };

    explicit scene(typename cloud_t::Ptr cloud, int device = 0) : scene(cloud, detail::default_context(device)) {}
    /* on a context of the caller's (make_context; the model it is searched against lives on the same one) */
    scene(typename cloud_t::Ptr cloud, context_t context) : cloud_(cloud), ctx_(std::move(context)) {
        static_assert(sizeof(Point) == 48, "Point must be the 48-byte surfel layout");
        if (!ctx_) throw std::invalid_argument("scene: null context");
        if (!cloud_ || cloud_->size() == 0) throw std::runtime_error("scene: empty cloud");
    }
    virtual ~scene() { tm_scene_destroy(h_); }
    scene(const scene&) = delete;
    scene& operator=(const scene&) = delete;

    /* the curvature-ratio half of tangent_mask_ (include/impl/scene.hpp:50) is a "next" row; supply one
     * flag per point to apply it */
    void set_curvature_ok(const std::vector<uint8_t>& ok) { curvature_ok_ = ok; }
    /* curvature half of tangent_mask_ (kNN-k, pc_min / pc_max < 0.2, include/impl/scene.hpp:49-52): on by
     * default with the reference's k = 30; disable_curvature_filter() opts out */
    void enable_curvature_filter(uint32_t k = 30) { curvature_k_ = k; }
    void disable_curvature_filter() { curvature_k_ = 0; }
    /* Multi-GPU search: one process per GPU, every process holds the same scene and model and calls
     * find_all_parallel with the same arguments; the first-point chunks of every round are dealt to the ranks
     * and one RCCL all-gather of 96-byte records per round brings their candidates together (search.hpp).
     * Every rank returns the same matches, and the same matches as a single process. */
    void set_communicator(tm_comm* comm) { comm_ = comm; }
    /* the device copy of the scene is kept between find_all_parallel calls; call this after changing the
     * cloud's points in place */
    void invalidate() {
        if (h_) tm_scene_destroy(h_);
        h_ = nullptr;
    }
    void set_workers(uint32_t w) { workers_ = w; }
    /* detail::early_drop of the reference (include/impl/scene.hpp:25) is true; switching it off scores
     * every hypothesis to the end and enables the batched GPU path (16 first points per launch) */
    void set_early_drop(bool on) { early_drop_ = on; }
    void set_verbose(bool on) { verbose_ = on; }
    /* wall-clock seconds per phase of the last find_all_parallel (search::stats_t) */
    const search::stats_t& last_stats() const { return stats_; }

    /* include/impl/scene.hpp:39-98 */
    std::vector<match_t> find_all_parallel(model<Point>& m, float dist_thres, float model_match_factor,
                                           float early_out_factor, const sample_parameters& sample_params,
                                           uint32_t max_icp_iterations) {
        /* The device copy is valid for the content it was made from (a hash of the cloud's bytes, ensure_scene). A
         * second search over the same scene object does not wait 0.6 ms per 48 MB for that hash: it searches the
         * device copy while the pool's idle workers hash the cloud, and compares when the search is over. Unequal —
         * the points were edited in place since the last call — what was found belongs to the old points: the scene is
         * prepared again and searched again. TM_HASH_FIRST=1: always hash before searching. */
        if (h_ && cached_n_ == (uint32_t)cloud_->size() && cached_curv_k_ == curvature_k_ && cached_curv_ok_ == curvature_ok_ &&
            !detail::env_once().hash_first) {
            typename cloud_t::hash_job hj;
            cloud_->content_hash_begin(hj);
            std::vector<match_t> results;
            std::exception_ptr failed;
            try {
                results = search_prepared(m, dist_thres, model_match_factor, early_out_factor, sample_params, max_icp_iterations);
            } catch (...) {
                failed = std::current_exception();
            }
            const uint64_t hash = cloud_->content_hash_end(hj);
            if (hash == cached_hash_) {
                if (failed) std::rethrow_exception(failed);
                return results;
            }
            ensure_scene(&hash);
            return search_prepared(m, dist_thres, model_match_factor, early_out_factor, sample_params, max_icp_iterations);
        }
        ensure_scene();
        return search_prepared(m, dist_thres, model_match_factor, early_out_factor, sample_params, max_icp_iterations);
    }

protected:
    /* the identity search over the prepared device scene (h_, cached_tmask_) */
    std::vector<match_t> search_prepared(model<Point>& m, float dist_thres, float model_match_factor, float early_out_factor,
                                         const sample_parameters& sample_params, uint32_t max_icp_iterations) {
        const std::vector<int32_t>& tangent_mask = cached_tmask_; /* tangent_mask_ (:46-54), kept with the device scene */
        detail::hip_backend<Point> be;
        be.s = h_;
        be.m = m.handle();
        be.scloud = cloud_.get();
        be.mcloud = m.cloud().get();
        be.n_t = m.point_count();
        be.diam = m.diameter();
        be.raw_rows = tm_model_get_feature_bounds(be.m, be.bmin, be.bmax) == TM_OK; /* the device scene = this cloud's rows */
        be.device_arbitration = true; /* tangent_mask below is the one the device scene was made with */
        stats_ = search::stats_t();
        fill_params(be.p, dist_thres, model_match_factor, early_out_factor, sample_params, max_icp_iterations);
        std::vector<search::candidate_t> found;
        /* the backend's staging vectors (correspondence lists of a round's candidates, neighbourhood views: a few MB)
         * outlive the call: a search does not allocate and zero them again */
        struct lend { /* ... and go back whichever way the search ends */
            detail::hip_backend<Point>& be;
            scene& sc;
            lend(detail::hip_backend<Point>& b, scene& s) : be(b), sc(s) { swap(); be.batch_stride = sc.keep_stride_; }
            ~lend() { swap(); sc.keep_stride_ = be.batch_stride; }
            void swap() {
                be.sc_scratch.swap(sc.keep_sc_);
                be.mc_scratch.swap(sc.keep_mc_);
                be.t_pairs.swap(sc.keep_pairs_);
            }
        };
        {
            lend lent(be, *this);
            if (comm_ && tm_comm_world(comm_) > 1) {
                detail::hip_exchange hx{comm_};
                found = search::find_all(be, tangent_mask, be.p, hx, nullptr, &keep_ws_);
            } else {
                search::single_rank one;
                found = search::find_all(be, tangent_mask, be.p, one, nullptr, &keep_ws_);
            }
        }
        std::vector<match_t> results;
        for (auto& c : found) {
            match_t r;
            memcpy(r.transform.m, c.transform.m, 64);
            r.transform = r.transform.inverse(); /* :92: model -> scene */
            r.scene_corrs = std::move(c.scene_corrs);
            r.model_corrs = std::move(c.model_corrs);
            r.signed_score = c.signed_score;
            results.push_back(std::move(r));
        }
        return results;
    }

public:
    /* One match of a projected search: `match` lives in uvw space (transform = uvw model -> uvw scene,
     * scene_corrs = indices into this scene's cloud, model_corrs = indices into the ORIGINAL model cloud);
     * `projector` is the scene projector it was found under. A model point x appears in the scene at
     * Traits::unproject(projector, match.transform * Traits::project(model projector, x)). */
    template <typename Traits>
    struct projected_match_t {
        match_t match;
        typename Traits::const_handle_t projector;
    };

    /* The traits search (projected_search.hpp): projector candidates from Traits::sample_count scene points,
     * Traits::init_from_samples, the scene mapped into uvw space on the device, the identity search there,
     * masking, next projector. max_icp_iterations refines in uvw space with a rigid fit; leave it 0 for the
     * cylinder (its u coordinate wraps around). */
    template <typename Traits>
    std::vector<projected_match_t<Traits>> find_all_parallel(projected_model<Traits, Point>& pm, float dist_thres,
                                                             float model_match_factor, float early_out_factor,
                                                             const sample_parameters& sample_params,
                                                             uint32_t max_icp_iterations,
                                                             const projector_search& ps = projector_search()) {
        std::vector<int32_t> base_tmask;
        const auto t_call0 = std::chrono::steady_clock::now();
        double ms_prepare = 0, ms_draw = 0, ms_probe = 0, ms_project = 0, ms_search = 0;
        auto since = [](std::chrono::steady_clock::time_point t0) {
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        };
        ensure_scene(base_tmask);
        const uint32_t n = (uint32_t)cloud_->size();
        model<Point>& um = pm.uvw_model();
        const uint32_t n_model = (uint32_t)um.cloud()->size();
        std::vector<int32_t> mask(n, 0);
        detail::check(tm_scene_set_mask(h_, mask.data()), "tm_scene_set_mask");
        std::vector<uint32_t> free_pts; /* finite points no accepted match holds */
        for (uint32_t i = 0; i < n; ++i)
            if (cloud_->points[i].getVector3fMap().allFinite() && cloud_->points[i].getNormalVector3fMap().allFinite())
                free_pts.push_back(i);
        std::mt19937 rng((uint32_t)ps.seed);
        std::vector<int32_t> nb; /* neighbourhood of the first sample */
        ms_prepare = since(t_call0);
        const uint32_t min_tangent = (uint32_t)(model_match_factor * (float)um.point_count()); /* = min_points, :114 */
        std::vector<projected_match_t<Traits>> results;
        std::vector<typename Traits::const_handle_t> searched_; /* projectors that had a full search */
        uint32_t skipped_surfaces_ = 0;
        uint32_t searches = 0, idle = 0;
        bool mask_dirty = false; /* the device mask was set to the (empty) mask above */
        stats_ = search::stats_t();
        uint32_t idle_draws = 0;
        for (uint32_t draw = 0; draw < ps.max_draws && searches < ps.max_searches && idle < ps.max_idle_searches &&
                                idle_draws < ps.max_idle_draws; ++draw, ++idle_draws) {
            if (free_pts.size() < (size_t)Traits::sample_count) break;
            /* the first sample anywhere among the unexplained points, the others in its neighbourhood
             * (radius = the model's diameter, the radius_search_inclusive of the search itself, :273): three
             * independent points of a room almost never share a wall, three neighbours often do */
            uint32_t idx[Traits::sample_count];
            bool ok = true;
            const auto t_d0 = std::chrono::steady_clock::now();
            struct add_on_exit { /* whatever way this draw ends, its time is booked */
                double& acc;
                std::chrono::steady_clock::time_point t0;
                bool armed = true;
                ~add_on_exit() {
                    if (armed) acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                }
            } draw_timer{ms_draw, t_d0};
            idx[0] = free_pts[rng() % (uint32_t)free_pts.size()];
            if (Traits::sample_count > 1) {
                /* only the size of the neighbourhood and the entries the draws pick are needed, not the list */
                uint32_t cnt = 0;
                detail::check(tm_scene_radius_subset(h_, idx[0], um.diameter(), nullptr, 0, &cnt), "tm_scene_radius_subset");
                if (cnt < (uint32_t)Traits::sample_count) continue;
                uint32_t ranks[Traits::sample_count];
                int32_t picked[Traits::sample_count];
                for (int k = 1; k < Traits::sample_count; ++k) ranks[k - 1] = rng() % cnt;
                detail::check(tm_scene_radius_select(h_, ranks, (uint32_t)(Traits::sample_count - 1), picked),
                              "tm_scene_radius_select");
                for (int k = 1; k < Traits::sample_count; ++k) {
                    idx[k] = (uint32_t)picked[k - 1];
                    for (int q = 0; q < k; ++q) ok = ok && idx[q] != idx[k];
                    ok = ok && !mask[idx[k]] && cloud_->points[idx[k]].getNormalVector3fMap().allFinite();
                }
            }
            if (!ok) continue;
            float flat[12 * Traits::sample_count], ms[TM_STATE_FLOATS] = {0}, st[TM_STATE_FLOATS] = {0};
            for (int k = 0; k < Traits::sample_count; ++k) memcpy(flat + 12 * k, cloud_->points[idx[k]].raw(), 48);
            pm.handle()->pack(ms);
            if (tm_traits_init_from_samples(Traits::kind, ms, flat, st)) continue; /* nullptr in the reference */
            auto handle = std::make_shared<typename Traits::state_t>();
            handle->unpack(st);
            if (Traits::kind == TM_TRAITS_CYLINDER &&
                !(std::fabs(handle->radius - pm.handle()->radius) <= ps.radius_tolerance * pm.handle()->radius))
                continue;
            if (ps.skip_searched_surfaces) { /* a surface that has had its search already? */
                const bool use_band = pm.handle()->threshold > 0.f;
                const float band = Traits::kind == TM_TRAITS_CYLINDER && pm.handle()->radius > 0.f
                                       ? pm.handle()->threshold / pm.handle()->radius
                                       : pm.handle()->threshold;
                bool seen = false;
                for (const auto& old : searched_) {
                    float dotp = 0.f; /* third rows of g2l: the plane's normal / the cylinder's axis */
                    for (int k = 0; k < 3; ++k) dotp += old->g2l(2, k) * handle->g2l(2, k);
                    if (!(std::fabs(dotp) > 0.9986f)) continue;
                    bool on = use_band;
                    for (int k = 0; k < Traits::sample_count && on; ++k) {
                        auto uvw = Traits::project(old, cloud_->points[idx[k]].getVector3fMap());
                        on = uvw && std::fabs((*uvw)[2]) <= band;
                    }
                    if (on) {
                        seen = true;
                        break;
                    }
                }
                if (seen) {
                    ++skipped_surfaces_;
                    continue;
                }
            }
            tm_scene* proj = nullptr;
            ms_draw += since(t_d0);
            draw_timer.armed = false;
            const auto t_p0 = std::chrono::steady_clock::now();
            if (mask_dirty) { /* mask_ changes only when a match was accepted */
                detail::check(tm_scene_set_mask(h_, mask.data()), "tm_scene_set_mask");
                mask_dirty = false;
            }
            {   /* the two rejections below, counted on the device before anything is built or downloaded */
                const float band = !(ps.on_surface_samples && pm.handle()->threshold > 0.f) ? -1.f
                                   : Traits::kind == TM_TRAITS_CYLINDER && pm.handle()->radius > 0.f
                                       ? pm.handle()->threshold / pm.handle()->radius
                                       : pm.handle()->threshold;
                uint64_t live = 0;
                uint32_t n_tan = 0;
                detail::check(tm_scene_probe_projected(h_, Traits::kind, st, band, &live, &n_tan), "tm_scene_probe_projected");
                ms_probe += since(t_p0);
                if ((float)live < model_match_factor * (float)n_model || n_tan < min_tangent) continue;
            }
            const auto t_c0 = std::chrono::steady_clock::now();
            struct add_project {
                double& acc;
                std::chrono::steady_clock::time_point t0;
                bool armed = true;
                ~add_project() {
                    if (armed) acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                }
            } project_timer{ms_project, t_c0};
            detail::check(tm_scene_create_projected(h_, Traits::kind, st, &proj), "tm_scene_create_projected");
            struct guard {
                tm_scene* s;
                ~guard() { tm_scene_destroy(s); }
            } g{proj};
            if ((float)tm_scene_live_count(proj) < model_match_factor * (float)n_model) continue;
            /* What the host side of the search reads of the uvw scene: its tangent points (first and second samples,
             * include/impl/scene.hpp:266-302) — the projected scene keeps the other points on the device. */
            std::vector<int32_t> tmask(n, 0);
            {
                uint32_t cnt = 0;
                detail::check(tm_scene_tangent_points(proj, nullptr, nullptr, 0, &cnt), "tm_scene_tangent_points");
                tan_idx_.resize(cnt);
                tan_rows_.resize(12 * (size_t)cnt);
                if (cnt)
                    detail::check(tm_scene_tangent_points(proj, tan_idx_.data(), tan_rows_.data(), cnt, &cnt),
                                  "tm_scene_tangent_points");
                const bool use_band = ps.on_surface_samples && pm.handle()->threshold > 0.f;
                const float band = Traits::kind == TM_TRAITS_CYLINDER && pm.handle()->radius > 0.f
                                       ? pm.handle()->threshold / pm.handle()->radius /* w = height / radius */
                                       : pm.handle()->threshold;
                uint32_t n_tan = 0; /* a surface with fewer tangent points than the arbitration demands as inliers
                                       cannot hold a match */
                for (uint32_t k = 0; k < cnt; ++k) {
                    const uint32_t i = tan_idx_[k];
                    if (use_band && !(std::fabs(tan_rows_[12 * (size_t)k + 2]) <= band)) continue;
                    tmask[i] = 1;
                    n_tan += mask[i] ? 0u : 1u;
                }
                if (n_tan < min_tangent) continue;
            }
            ++searches;
            idle_draws = 0;
            searched_.push_back(handle);
            ms_project += since(t_c0);
            project_timer.armed = false;
            const auto t_p1 = std::chrono::steady_clock::now();
            detail::hip_backend<Point> be;
            be.s = proj;
            be.m = um.handle();
            be.scloud = cloud_.get(); /* the size is all the search takes from it; the uvw points stay on the device */
            be.mcloud = um.cloud().get();
            be.n_t = um.point_count();
            be.diam = um.diameter();
            be.traits_kind = Traits::kind;
            fill_params(be.p, dist_thres, model_match_factor, early_out_factor, sample_params, max_icp_iterations);
            be.p.min_tangent_inliers = min_tangent;
            std::vector<search::candidate_t> found;
            if (comm_ && tm_comm_world(comm_) > 1) {
                detail::hip_exchange hx{comm_};
                found = search::find_all(be, tmask, be.p, hx, &mask);
            } else {
                search::single_rank one;
                found = search::find_all(be, tmask, be.p, one, &mask);
            }
            ms_search += since(t_p1);
            if (ps.verbose) {
                const auto t_p2 = std::chrono::steady_clock::now();
                fprintf(stderr, "[projected search] draw %u: projector searched (%llu points project), %zu match(es); "
                                "project + download %.1f ms, search %.1f ms (radius %.1f sample %.1f generate+score %.1f "
                                "extract %.1f icp %.1f arbitration %.1f)\n", draw,
                        (unsigned long long)tm_scene_live_count(proj), found.size(),
                        std::chrono::duration<double, std::milli>(t_p1 - t_p0).count(),
                        std::chrono::duration<double, std::milli>(t_p2 - t_p1).count(), stats_.radius * 1e3,
                        stats_.sample * 1e3, stats_.generate_score * 1e3, stats_.extract * 1e3, stats_.icp * 1e3,
                        stats_.arbitration * 1e3);
            }
            if (found.empty()) {
                ++idle;
                continue;
            }
            idle = 0;
            for (auto& c : found) {
                projected_match_t<Traits> r;
                memcpy(r.match.transform.m, c.transform.m, 64);
                r.match.transform = r.match.transform.inverse(); /* uvw model -> uvw scene, as :92 */
                r.match.signed_score = c.signed_score;
                for (uint32_t i : c.scene_corrs) mask[i] = 1;
                mask_dirty = true;
                r.match.scene_corrs = std::move(c.scene_corrs);
                r.match.model_corrs.reserve(c.model_corrs.size());
                for (uint32_t k : c.model_corrs) r.match.model_corrs.push_back(pm.original_index(k));
                r.projector = handle;
                results.push_back(std::move(r));
            }
            size_t w = 0;
            for (uint32_t i : free_pts)
                if (!mask[i]) free_pts[w++] = i;
            free_pts.resize(w);
        }
        last_projector_searches_ = searches;
        if (ps.verbose)
            fprintf(stderr, "[projected search] %.1f ms in all: prepare the scene %.1f, draws + init_from_samples %.1f, probes %.1f, "
                            "project + prepare %.1f, searches %.1f (%u searched, %u sample sets on surfaces already searched)\n",
                    since(t_call0), ms_prepare, ms_draw, ms_probe, ms_project, ms_search, searches, skipped_surfaces_);
        return results;
    }
    /* projectors that got a full search in the last projected find_all_parallel */
    uint32_t last_projector_searches() const { return last_projector_searches_; }

    /* include/impl/scene.hpp:512-531 (instantiate_all == false returns an empty cloud there too) */
    typename cloud_t::Ptr instantiate(model<Point>& m, const match_t& match, bool instantiate_all = false) {
        typename cloud_t::Ptr inst = cloud_t::empty();
        if (!instantiate_all) return inst;
        for (const auto& mp : m.cloud()->points) {
            vec4f_t q = match.transform * homogeneous(mp.getVector3fMap());
            Point o;
            o.x = q[0];
            o.y = q[1];
            o.z = q[2];
            inst->push_back(o);
        }
        return inst;
    }

    typename cloud_t::ConstPtr cloud() const { return cloud_; }
    tm_scene* handle() const { return h_; }

protected:
    /* tangent_mask_ (include/impl/scene.hpp:46-54) and the device copy of the scene: the prepared scene (56 MB
     * upload + sort for 1M points) is reused while the cloud's CONTENT (a 64-bit hash of the raw surfel bytes,
     * ~5 ms per million points) and the mask settings are unchanged — the reference re-reads cloud_ on every call,
     * so an in-place edit of the points between two calls must be seen; the searches reset mask_ themselves */
    /* a copy of tangent_mask_ for a caller that edits it (the projected searches) */
    void ensure_scene(std::vector<int32_t>& tangent_mask) {
        ensure_scene(nullptr);
        tangent_mask = cached_tmask_;
    }
    /* the device scene and tangent_mask_ (cached_tmask_) for the cloud's current content */
    void ensure_scene(const uint64_t* known_hash = nullptr) {
        const uint32_t n = (uint32_t)cloud_->size();
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) { /* set_verbose(true): where the scene preparation's time goes */
            if (!verbose_) return;
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[scene] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
            t_last = now;
        };
        const uint64_t hash = known_hash ? *known_hash : cloud_->content_hash();
        lap("content hash");
        cloud_->sync(hash); /* curvature computed for other content is forgotten */
        if (h_ && hash != cached_hash_) invalidate();
        if (h_ && cached_n_ == n && cached_curv_k_ == curvature_k_ && cached_curv_ok_ == curvature_ok_)
            return; /* same content, same settings: the mask is the one already built */
        std::vector<int32_t> tangent_mask(n, 0);
        keep_ws_.tangent_indices_valid = false; /* the kept list of tangent points belongs to the old mask */
        if (!detail::env_once().host_curvature_tree) {
            /* tangent_mask_ (:46-54) where the cloud is: the scene goes up with an empty tangent mask, and one call
             * finds the points with |tangent| > 0.7 (and the caller's flag), computes the kNN curvature of exactly
             * those on a tree built over the scene's curve order, applies pc_min / pc_max < 0.2 and makes the result
             * the scene's mask (tm_scene_auto_tangent_mask) — no loop over the cloud on this side (1.8 + 3.4 ms per
             * million points), no host tree. The candidates' rows are kept as the cloud's partial curvature cache. */
            if (h_ && cached_n_ != n) invalidate();
            if (!h_) {
                detail::check(tm_scene_create(ctx_->ctx, cloud_->points[0].raw(), n, nullptr, &h_), "tm_scene_create");
                lap("tm_scene_create");
            }
            std::vector<uint32_t> cand(n / 8 + 1024);
            std::vector<float> rows(5 * cand.size());
            uint32_t n_cand = 0, n_tan = 0;
            const unsigned char* ok = curvature_ok_.empty() ? nullptr : curvature_ok_.data();
            int rc = tm_scene_auto_tangent_mask(h_, curvature_k_, 0.7f, 0.2f, ok, tangent_mask.data(), cand.data(), rows.data(),
                                                (uint32_t)cand.size(), &n_cand, &n_tan);
            if (rc == TM_ERR_CAPACITY) { /* more candidates than a good eighth of the cloud: once more with room for all */
                cand.resize(n_cand);
                rows.resize(5 * (size_t)n_cand);
                rc = tm_scene_auto_tangent_mask(h_, curvature_k_, 0.7f, 0.2f, ok, tangent_mask.data(), cand.data(), rows.data(),
                                                n_cand, &n_cand, &n_tan);
            }
            detail::check(rc, "tm_scene_auto_tangent_mask");
            cand.resize(n_cand);
            rows.resize(5 * (size_t)n_cand);
            /* the tangent points are among the candidates of the |tangent| test (ascending): the search's list of them
             * comes from those few per cent of the cloud instead of a pass over the whole mask */
            keep_ws_.tangent_indices.clear();
            for (uint32_t i : cand)
                if (tangent_mask[i]) keep_ws_.tangent_indices.push_back(i);
            keep_ws_.tangent_indices_valid = keep_ws_.tangent_indices.size() == n_tan;
            if (curvature_k_) cloud_->store_curvature(curvature_k_, std::move(cand), std::move(rows), hash);
            lap("tangent_mask_ on the device");
        } else {
            /* TM_HOST_CURVATURE_TREE=1 (cross-check): the mask loop of :46-54 on this side over curvature from a tree
             * built on the host */
            if (curvature_k_) {
                std::vector<uint32_t> cand;
                for (uint32_t i = 0; i < n; ++i)
                    if (tangent(cloud_->points[i]).norm() > 0.7f && (curvature_ok_.empty() || curvature_ok_[i])) cand.push_back(i);
                lap("candidates of the tangent test");
                cloud_->prepare_curvature(curvature_k_, cand);
                lap("curvature of the candidates");
            }
            for (uint32_t i = 0; i < n; ++i) {
                bool t = tangent(cloud_->points[i]).norm() > 0.7f; /* :50 */
                if (t && !curvature_ok_.empty() && !curvature_ok_[i]) t = false;
                if (t && curvature_k_) {
                    curvature_info_t ci = cloud_->curvature(curvature_k_, i); /* :49-50 */
                    if (!((ci.pc_min / ci.pc_max) < 0.2f)) t = false;
                }
                tangent_mask[i] = t ? 1 : 0;
            }
            lap("tangent_mask_");
            if (h_ && cached_n_ != n) invalidate();
            if (!h_) {
                detail::check(tm_scene_create(ctx_->ctx, cloud_->points[0].raw(), n, tangent_mask.data(), &h_),
                              "tm_scene_create");
            } else if (cached_tmask_ != tangent_mask) { /* same cloud on the device, another tangent_mask_ */
                detail::check(tm_scene_set_tangent_mask(h_, tangent_mask.data()), "tm_scene_set_tangent_mask");
            }
            lap("scene on the device");
        }
        cached_n_ = n;
        cached_tmask_.swap(tangent_mask);
        cached_hash_ = hash;
        cached_curv_k_ = curvature_k_;
        cached_curv_ok_ = curvature_ok_;
    }
    void fill_params(search::params_t& p, float dist_thres, float model_match_factor, float early_out_factor,
                     const sample_parameters& sample_params, uint32_t max_icp_iterations) {
        p.dist_thres = dist_thres;
        p.accept_prob = model_match_factor;
        p.early_out_factor = early_out_factor;
        p.min_diameter_factor = sample_params.min_diameter_factor;
        p.max_diameter_factor = sample_params.max_diameter_factor;
        p.force_up = sample_params.force_up;
        p.max_icp_iterations = max_icp_iterations;
        p.workers = workers_;
        p.early_drop = early_drop_;
        p.verbose = verbose_;
        p.stats = &stats_;
    }

    typename cloud_t::Ptr cloud_;
    std::shared_ptr<detail::context_holder> ctx_;
    tm_scene* h_ = nullptr;
    std::vector<uint32_t> keep_sc_, keep_mc_, keep_pairs_; /* hip_backend staging kept between searches */
    size_t keep_stride_ = 0;                                /* ... and the rows per pose its batched finish_find needed */
    search::find_all_workspace keep_ws_;                    /* ... and find_all's masks and list of tangent points */
    uint32_t last_projector_searches_ = 0;
    std::vector<uint32_t> tan_idx_; /* tangent points of the current projected scene */
    std::vector<float> tan_rows_;
    std::vector<uint8_t> curvature_ok_;
    uint32_t workers_ = 7;
    uint32_t curvature_k_ = 30;
    tm_comm* comm_ = nullptr;
    uint32_t cached_n_ = 0;
    std::vector<int32_t> cached_tmask_;
    uint64_t cached_hash_ = 0;
    uint32_t cached_curv_k_ = 0;
    std::vector<uint8_t> cached_curv_ok_;
    bool early_drop_ = true;
    bool verbose_ = false;
    search::stats_t stats_;
};

}  // namespace triplet_match

#endif
