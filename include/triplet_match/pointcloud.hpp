/*
 * triplet_match/pointcloud.hpp — cloud container of the drop-in API.
 *
 * Mirrors the parts of the reference's pointcloud<Point> (include/pointcloud:13-93) the hot path
 * consumes: points, size(), from_pcd() (include/impl/pointcloud.hpp:58-64), resolution() (mean 1-NN
 * distance, :66-82; tm_cloud_resolution, GPU kd-tree 2-NN, unless set_resolution() supplied a value),
 * diameter() (:84-97), curvature(k, idx) (:200-204; tm_cloud_curvature), and the neighbour accessors
 * knn_inclusive / knn_exclusive / radius_search_inclusive / radius_search_exclusive for an index or a point,
 * curvature(float r, idx), set_indices(), reset_tree() (include/pointcloud:38-63, impl :100-210) over a tm_cloud
 * handle (kd-tree on the device, built lazily under a mutex like the reference's tree). Result order =
 * (squared distance, index) ascending, distances squared (as PCL returns). Not provided: kdtree() / native()
 * (they hand out PCL objects) and the range-v3 view adaptors (unused; one of them does not compile upstream).
 */
#ifndef TRIPLET_MATCH_POINTCLOUD_HPP_
#define TRIPLET_MATCH_POINTCLOUD_HPP_

#include <algorithm>
#include <limits>
#include <mutex>
#include <thread>
#include <utility>

#include "common.hpp"
#include "pcd_io.hpp"
#include "worker_pool.hpp"

namespace triplet_match {

struct curvature_info_t { /* include/pointcloud:24-28 */
    vec3f_t pc_max_dir;
    float pc_min = 0.f, pc_max = 0.f;
};

template <typename Point>
class pointcloud {
public:
    typedef std::shared_ptr<pointcloud<Point>> Ptr;
    typedef std::shared_ptr<const pointcloud<Point>> ConstPtr;

    std::vector<Point> points;

    static Ptr empty() { return std::make_shared<pointcloud<Point>>(); }

    /* include/impl/pointcloud.hpp:58-64. The reference ignores loadPCDFile's status, so a missing or
     * malformed file yields an empty cloud there; here the reason is also printed. pcd::read throws. */
    static Ptr from_pcd(const std::string& filename) {
        Ptr self = empty();
        try {
            pcd::read(filename, self->points);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "[triplet_match::pointcloud::from_pcd] %s\n", e.what());
            self->points.clear();
        }
        return self;
    }

    pointcloud() = default;
    pointcloud(const pointcloud& o) : points(o.points), subset_(o.subset_) {}
    pointcloud& operator=(const pointcloud& o) {
        if (this != &o) {
            points = o.points;
            subset_ = o.subset_;
            touch();
        }
        return *this;
    }
    ~pointcloud() { tm_cloud_destroy(tree_); }

    size_t size() const { return points.size(); }
    void push_back(const Point& p) {
        points.push_back(p);
        touch();
    }
    /* Call after changing `points` in place: forgets resolution, diameter, curvature and the search tree. (The
     * reference recomputes nothing either — its caches are just as sticky — but scene<Point> re-reads the cloud on
     * every find_all_parallel; content_hash() below is how our scene notices an edit.) */
    void touch() {
        std::lock_guard<std::mutex> lk(mutex_);
        if (!resolution_fixed_) resolution_ = -1.f;
        diameter_ = -1.f;
        {
            std::lock_guard<std::mutex> lc(curv_mutex_);
            curv_k_ = 0;
        }
        tm_cloud_destroy(tree_);
        tree_ = nullptr;
    }
    /* forgets the caches if they were filled for other content than `hash` (= content_hash() of now) */
    void sync(uint64_t hash) {
        bool stale;
        {
            std::lock_guard<std::mutex> lc(curv_mutex_);
            stale = curv_k_ != 0 && curv_hash_ != hash;
        }
        if (stale) touch();
    }
    /* 64-bit hash of the raw surfel bytes (eight interleaved multiply-xorshift lanes per slice of 128k words, slices on
     * the pool, combined in slice order): scene<Point> keys its device copy on it, so an in-place edit of the points is
     * never searched stale */
    static constexpr size_t kHashSlice = (size_t)1 << 17;
    size_t hash_slices() const {
        const size_t n_words = points.size() * (sizeof(Point) / 8);
        return std::max<size_t>(1, (n_words + kHashSlice - 1) / kHashSlice);
    }
    uint64_t hash_slice(size_t sl) const {
        const size_t n_words = points.size() * (sizeof(Point) / 8);
        const uint64_t* w = reinterpret_cast<const uint64_t*>(points.data());
        const size_t lo = sl * kHashSlice, hi = std::min(n_words, lo + kHashSlice);
        /* eight independent lanes: a lane's xor - multiply - shift - xor chain is ~6 cycles per word, so four
         * lanes left the multiplier idle half of the time (0.74 -> 0.54 ms per 48 MB on the pool's threads) */
        constexpr int kLanes = 8;
        uint64_t h[kLanes] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull,
                              0x85EBCA77C2B2AE63ull, 0xD6E8FEB86659FD93ull, 0xA0761D6478BD642Full, 0xE7037ED1A0B428DBull};
        size_t i = lo;
        for (; i + kLanes <= hi; i += kLanes)
            for (int l = 0; l < kLanes; ++l) {
                uint64_t v;
                std::memcpy(&v, w + i + l, 8);
                h[l] = (h[l] ^ v) * 0xFF51AFD7ED558CCDull;
                h[l] ^= h[l] >> 29;
            }
        for (; i < hi; ++i) {
            uint64_t v;
            std::memcpy(&v, w + i, 8);
            h[0] = (h[0] ^ v) * 0xFF51AFD7ED558CCDull;
            h[0] ^= h[0] >> 29;
        }
        uint64_t r = (uint64_t)(hi > lo ? hi - lo : 0);
        for (int l = 0; l < kLanes; ++l) {
            r = (r ^ h[l]) * 0xC4CEB9FE1A85EC53ull;
            r ^= r >> 32;
        }
        return r;
    }
    uint64_t hash_combine(const std::vector<uint64_t>& part) const {
        uint64_t r = (uint64_t)points.size();
        for (uint64_t pv : part) {
            r = (r ^ pv) * 0xC4CEB9FE1A85EC53ull;
            r ^= r >> 32;
        }
        return r;
    }
    uint64_t content_hash() const {
        std::vector<uint64_t> part(hash_slices(), 0);
        /* on the headers' worker pool (threads started per call cost a third of the 0.9 ms) */
        detail::worker_pool::instance().run(part.size(), [&](size_t sl) { part[sl] = hash_slice(sl); });
        return hash_combine(part);
    }
    /* The same hash as a background job of the pool: begun before a search that assumes the content unchanged, read
     * when the search is over (scene<Point>::find_all_parallel) — the 0.6 ms per 48 MB leave the search's critical
     * path. The points must not change between the two calls. */
    struct hash_job {
        std::vector<uint64_t> part;
        detail::worker_pool::job job;
    };
    void content_hash_begin(hash_job& hj) const {
        hj.part.assign(hash_slices(), 0);
        uint64_t* part = hj.part.data();
        detail::worker_pool::instance().begin_background(hj.job, hj.part.size(), [this, part](size_t sl) { part[sl] = hash_slice(sl); });
    }
    uint64_t content_hash_end(hash_job& hj) const {
        detail::worker_pool::instance().finish(hj.job);
        return hash_combine(hj.part);
    }

    void set_resolution(float r) { /* a value supplied by the caller survives touch() */
        resolution_ = r;
        resolution_fixed_ = true;
    }

    /* mean distance to the nearest other point (include/impl/pointcloud.hpp:66-82): 2-NN distances from
     * a kd-tree kernel on the GPU, running mean accum + (val - accum)/(++i) in f32 in point order as the
     * reference's average() does (include/common:104-115) */
    float resolution() const {
        if (resolution_ >= 0.f) return resolution_;
        static_assert(sizeof(Point) == 48, "Point must be the 48-byte surfel layout");
        if (points.size() < 2) return resolution_ = 0.f;
        auto ctx = detail::default_context(0);
        float r = 0.f;
        detail::check(tm_cloud_resolution(ctx->ctx, points[0].raw(), (uint32_t)points.size(), &r, nullptr),
                      "tm_cloud_resolution");
        return resolution_ = r;
    }

    /* curvature(k, idx), include/impl/pointcloud.hpp:200-204: kNN-k + principal curvatures of the projected
     * normals (tm_cloud_curvature, GPU); all points are computed on first use and cached per k. A partial cache
     * (store_curvature / prepare_curvature: the rows of the points a mask loop reads) answers for its points. */
    curvature_info_t curvature(uint32_t k, uint32_t idx) const {
        std::lock_guard<std::mutex> lc(curv_mutex_); /* scenes on several threads may share one cloud object */
        const float* row = nullptr;
        if (curv_k_ == k && curv_sparse_) {
            auto it = std::lower_bound(curv_idx_.begin(), curv_idx_.end(), idx);
            if (it != curv_idx_.end() && *it == idx) row = &curv_rows_[5 * (size_t)(it - curv_idx_.begin())];
        }
        if (!row) {
            if (!(curv_k_ == k && !curv_sparse_ && curv_.size() == 5 * points.size())) { /* nothing yet, or a partial cache without idx */
                curv_.assign(5 * points.size(), 0.f);
                auto ctx = detail::default_context(0);
                if (!points.empty())
                    detail::check(tm_cloud_curvature(ctx->ctx, points[0].raw(), (uint32_t)points.size(), k, curv_.data()),
                                  "tm_cloud_curvature");
                curv_sparse_ = false;
                curv_idx_.clear();
                curv_rows_.clear();
                curv_k_ = k;
                curv_hash_ = content_hash();
            }
            row = &curv_[5 * (size_t)idx];
        }
        curvature_info_t c;
        c.pc_max_dir = {row[0], row[1], row[2]};
        c.pc_min = row[3];
        c.pc_max = row[4];
        return c;
    }
    /* curvature(k, .) of the listed points in one go (tm_cloud_curvature_subset): what model::init and
     * scene::find_all_parallel ask before their mask loops — they read the curvature of the points that pass the
     * tangent test only (include/impl/model.hpp:96-99, scene.hpp:50), 4 % of a cloud, where the reference computes
     * all of them (:68-71, :48-49). A later curvature(k, idx) of another point recomputes the whole cloud.
     * `indices` ascending. */
    bool has_curvature(uint32_t k, const std::vector<uint32_t>& indices) const {
        std::lock_guard<std::mutex> lc(curv_mutex_);
        if (curv_k_ != k) return false;
        if (!curv_sparse_) return curv_.size() == 5 * points.size();
        size_t at = 0; /* both lists ascend: one pass */
        for (uint32_t i : indices) {
            while (at < curv_idx_.size() && curv_idx_[at] < i) ++at;
            if (at == curv_idx_.size() || curv_idx_[at] != i) return false;
        }
        return true;
    }
    /* rows computed elsewhere for this cloud's content (tm_scene_curvature_subset / tm_scene_auto_tangent_mask on a
     * scene's device copy of it); kept as (ascending index, row) pairs — a dense array cost 20 MB of stores per million
     * points for 40k rows. `hash` = content_hash() of the content they belong to. */
    void store_curvature(uint32_t k, std::vector<uint32_t> indices, std::vector<float> rows, uint64_t hash) const {
        std::lock_guard<std::mutex> lc(curv_mutex_);
        curv_idx_ = std::move(indices);
        curv_rows_ = std::move(rows);
        curv_sparse_ = true;
        curv_.clear();
        curv_.shrink_to_fit();
        curv_k_ = k;
        curv_hash_ = hash;
    }
    void prepare_curvature(uint32_t k, const std::vector<uint32_t>& indices) const {
        if (has_curvature(k, indices)) return;
        std::vector<float> rows(5 * indices.size());
        if (!indices.empty()) {
            auto ctx = detail::default_context(0);
            detail::check(tm_cloud_curvature_subset(ctx->ctx, points[0].raw(), (uint32_t)points.size(), k, indices.data(),
                                                    (uint32_t)indices.size(), rows.data()),
                          "tm_cloud_curvature_subset");
        }
        store_curvature(k, indices, std::move(rows), content_hash());
    }

    /* ---- neighbour accessors (include/pointcloud:38-63) ---- */
    typedef std::pair<std::vector<int>, std::vector<float>> neighbours_t; /* (indices, squared distances) */

    /* set_indices(subset): the tree is rebuilt over these points only (include/pointcloud:95-103, impl :224-240) */
    void set_indices(subset_t subset) {
        std::lock_guard<std::mutex> lk(mutex_);
        subset_ = std::move(subset);
        tm_cloud_destroy(tree_);
        tree_ = nullptr;
    }
    void reset_tree() { /* include/impl/pointcloud.hpp:105-111 */
        std::lock_guard<std::mutex> lk(mutex_);
        tm_cloud_destroy(tree_);
        tree_ = nullptr;
    }

    /* knn_inclusive(k, point), include/impl/pointcloud.hpp:138-146: the k nearest points of the tree, the query's
     * own point included when it is one of them */
    neighbours_t knn_inclusive(uint32_t k, const Point& point) const {
        neighbours_t out;
        if (k == 0) return out;
        tm_cloud* t = tree();
        const float q[3] = {point.x, point.y, point.z};
        out.first.assign(k, -1);
        out.second.assign(k, 0.f);
        uint32_t count = 0;
        detail::check(tm_cloud_knn(t, q, 1, k, out.first.data(), out.second.data(), &count), "tm_cloud_knn");
        out.first.resize(count); /* PCL resizes to what it found */
        out.second.resize(count);
        return out;
    }
    neighbours_t knn_inclusive(uint32_t k, uint32_t idx) const { return knn_inclusive(k, points[idx]); } /* :148-152 */
    /* knn_exclusive: k + 1 neighbours without the first (:154-166) */
    neighbours_t knn_exclusive(uint32_t k, const Point& point) const { return tail(knn_inclusive(k + 1, point)); }
    neighbours_t knn_exclusive(uint32_t k, uint32_t idx) const { return knn_exclusive(k, points[idx]); }
    /* several queries in one launch (no counterpart in the reference, whose callers loop): row-major k columns,
     * rows padded with -1 / +inf */
    void knn_batch(uint32_t k, const std::vector<uint32_t>& queries, std::vector<int>& indices,
                   std::vector<float>& sq_dists) const {
        indices.assign((size_t)queries.size() * k, -1);
        sq_dists.assign((size_t)queries.size() * k, 0.f);
        if (queries.empty() || k == 0) return;
        std::vector<float> q(3 * queries.size());
        for (size_t i = 0; i < queries.size(); ++i) {
            q[3 * i] = points[queries[i]].x;
            q[3 * i + 1] = points[queries[i]].y;
            q[3 * i + 2] = points[queries[i]].z;
        }
        detail::check(tm_cloud_knn(tree(), q.data(), (uint32_t)queries.size(), k, indices.data(), sq_dists.data(),
                                   nullptr),
                      "tm_cloud_knn");
    }

    /* radius_search_inclusive(r, point), :168-176 */
    neighbours_t radius_search_inclusive(float r, const Point& point) const {
        neighbours_t out;
        tm_cloud* t = tree();
        const float q[3] = {point.x, point.y, point.z};
        uint32_t count = 0;
        detail::check(tm_cloud_radius_search(t, q, r, nullptr, nullptr, 0, &count), "tm_cloud_radius_search");
        out.first.resize(count);
        out.second.resize(count);
        if (count)
            detail::check(tm_cloud_radius_search(t, q, r, out.first.data(), out.second.data(), count, &count),
                          "tm_cloud_radius_search");
        return out;
    }
    neighbours_t radius_search_inclusive(float r, uint32_t idx) const { return radius_search_inclusive(r, points[idx]); }
    neighbours_t radius_search_exclusive(float r, const Point& point) const { /* :184-190 */
        return tail(radius_search_inclusive(r, point));
    }
    neighbours_t radius_search_exclusive(float r, uint32_t idx) const { return radius_search_exclusive(r, points[idx]); }

    /* curvature(float r, idx), :206-209: principal curvatures over radius_search_inclusive(r, idx).first */
    curvature_info_t curvature(float r, uint32_t idx) const {
        const neighbours_t nb = radius_search_inclusive(r, idx);
        float c5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        detail::check(tm_cloud_curvature_of(tree(), idx, nb.first.data(), (uint32_t)nb.first.size(), c5),
                      "tm_cloud_curvature_of");
        curvature_info_t c;
        c.pc_max_dir = {c5[0], c5[1], c5[2]};
        c.pc_min = c5[3];
        c.pc_max = c5[4];
        return c;
    }

    float diameter() const { /* bounding-box diagonal, include/impl/pointcloud.hpp:84-97 */
        if (diameter_ >= 0.f) return diameter_;
        float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
                       std::numeric_limits<float>::max()};
        float mx[3] = {-mn[0], -mn[0], -mn[0]};
        for (const auto& p : points) {
            const float c[3] = {p.x, p.y, p.z};
            for (int d = 0; d < 3; ++d) {
                mn[d] = std::min(mn[d], c[d]);
                mx[d] = std::max(mx[d], c[d]);
            }
        }
        return diameter_ = vec3f_t(mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]).norm();
    }

private:
    static neighbours_t tail(neighbours_t nb) { /* vw::tail of both lists (:157-158) */
        if (!nb.first.empty()) {
            nb.first.erase(nb.first.begin());
            nb.second.erase(nb.second.begin());
        }
        return nb;
    }
    /* kdtree() const (:120-128): built on first use, under the mutex */
    tm_cloud* tree() const {
        std::lock_guard<std::mutex> lk(mutex_);
        if (!tree_) {
            static_assert(sizeof(Point) == 48, "Point must be the 48-byte surfel layout");
            auto ctx = detail::default_context(0);
            detail::check(tm_cloud_create(ctx->ctx, points.empty() ? nullptr : points[0].raw(), (uint32_t)points.size(),
                                          subset_.empty() ? nullptr : subset_.data(), (uint32_t)subset_.size(), &tree_),
                          "tm_cloud_create");
        }
        return tree_;
    }

    mutable tm_cloud* tree_ = nullptr;
    mutable std::mutex mutex_;
    subset_t subset_;
    mutable float resolution_ = -1.f;
    bool resolution_fixed_ = false;
    mutable float diameter_ = -1.f;
    mutable std::vector<float> curv_;         /* every point's row (dense cache) ... */
    mutable std::vector<uint32_t> curv_idx_;  /* ... or the rows of these points only, ascending (partial cache) */
    mutable std::vector<float> curv_rows_;
    mutable bool curv_sparse_ = false;
    mutable std::mutex curv_mutex_;           /* guards the curvature cache (mutex_ guards the tree and the scalars) */
    mutable uint32_t curv_k_ = 0;
    mutable uint64_t curv_hash_ = 0;
};

}  // namespace triplet_match

#endif
