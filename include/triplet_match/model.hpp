/*
 * triplet_match/model.hpp — model<Point> of the drop-in API over the C ABI.
 * Mirrors include/model:15-66 / include/impl/model.hpp: same constructor, init() overloads, query(),
 * voxel_query() and getters; the voxel NN grid, the pair enumeration and the key table live in HBM and
 * are built by HIP kernels (tm_model_build_voxels / tm_model_build_pairs).
 */
#ifndef TRIPLET_MATCH_MODEL_HPP_
#define TRIPLET_MATCH_MODEL_HPP_

#include <algorithm>
#include <iterator>
#include <mutex>

#include "feature.hpp"

namespace triplet_match {

constexpr int voxel_multiplicity = 1; /* include/model:13 */

namespace detail {
/* What query() hands out in place of the reference's unordered_multimap const_iterator (include/model:25-26):
 * a random-access iterator over one key's entries in the host mirror of the table; `it->second` is the (i, j)
 * tuple, `it->first` the key, as with the reference's nodes. It points into storage owned by the model that is
 * written once (after init) and never again, so ranges of concurrent queries never touch each other. */
template <typename Key, typename Result>
class table_iterator {
public:
    typedef std::pair<Key, Result> value_type;
    typedef value_type reference;
    typedef std::ptrdiff_t difference_type;
    typedef std::random_access_iterator_tag iterator_category;
    struct pointer { /* arrow proxy: keeps the synthesised entry alive for the `->` expression */
        value_type e;
        const value_type* operator->() const { return &e; }
    };
    table_iterator() = default;
    table_iterator(const uint32_t* ij, const Key& key) : p_(ij), key_(key) {}
    value_type operator*() const { return {key_, Result{p_[0], p_[1]}}; }
    pointer operator->() const { return pointer{**this}; }
    value_type operator[](difference_type k) const { return *(*this + k); }
    table_iterator& operator++() { p_ += 2; return *this; }
    table_iterator operator++(int) { table_iterator t = *this; p_ += 2; return t; }
    table_iterator& operator--() { p_ -= 2; return *this; }
    table_iterator operator--(int) { table_iterator t = *this; p_ -= 2; return t; }
    table_iterator& operator+=(difference_type k) { p_ += 2 * k; return *this; }
    table_iterator& operator-=(difference_type k) { p_ -= 2 * k; return *this; }
    friend table_iterator operator+(table_iterator a, difference_type k) { return a += k; }
    friend table_iterator operator+(difference_type k, table_iterator a) { return a += k; }
    friend table_iterator operator-(table_iterator a, difference_type k) { return a -= k; }
    friend difference_type operator-(const table_iterator& a, const table_iterator& b) { return (a.p_ - b.p_) / 2; }
    friend bool operator==(const table_iterator& a, const table_iterator& b) { return a.p_ == b.p_; }
    friend bool operator!=(const table_iterator& a, const table_iterator& b) { return a.p_ != b.p_; }
    friend bool operator<(const table_iterator& a, const table_iterator& b) { return a.p_ < b.p_; }
    friend bool operator>(const table_iterator& a, const table_iterator& b) { return a.p_ > b.p_; }
    friend bool operator<=(const table_iterator& a, const table_iterator& b) { return a.p_ <= b.p_; }
    friend bool operator>=(const table_iterator& a, const table_iterator& b) { return a.p_ >= b.p_; }

private:
    const uint32_t* p_ = nullptr;
    Key key_{};
};
}  // namespace detail

template <typename Point>
class scene;

template <typename Point>
class model {
public:
    typedef std::unique_ptr<model<Point>> uptr_t;
    typedef std::shared_ptr<model<Point>> sptr_t;
    typedef std::weak_ptr<model<Point>> wptr_t;
    typedef std::shared_ptr<const model<Point>> const_sptr_t;
    typedef std::weak_ptr<const model<Point>> const_wptr_t;
    typedef pointcloud<Point> cloud_t;
    typedef std::tuple<uint32_t, uint32_t> result_t;
    /* the reference iterates unordered_multimap nodes: `it->second` is the (i, j) tuple */
    typedef std::pair<discrete_feature_t, result_t> entry_t;
    typedef detail::table_iterator<discrete_feature_t, result_t> pair_iter_t;
    typedef std::array<uint32_t, voxel_multiplicity> voxel_data_t;

    model(typename cloud_t::ConstPtr cloud, discretization_params params, int device = 0)
        : model(cloud, params, detail::default_context(device)) {}
    /* on a context of the caller's (make_context): see common.hpp */
    model(typename cloud_t::ConstPtr cloud, discretization_params params, context_t context)
        : cloud_(cloud), params_(params), ctx_(std::move(context)) {
        static_assert(sizeof(Point) == 48, "Point must be the 48-byte surfel layout");
        if (!ctx_) throw std::invalid_argument("model: null context");
        if (!cloud_ || cloud_->size() == 0) throw std::runtime_error("model: empty cloud");
        detail::check(tm_model_create(ctx_->ctx, cloud_->points[0].raw(), (uint32_t)cloud_->size(),
                                      cloud_->resolution(), params_.distance_step_count, params_.angle_step,
                                      &h_),
                      "tm_model_create");
    }
    virtual ~model() { tm_model_destroy(h_); }
    model(const model&) = delete;
    model& operator=(const model&) = delete;

    /* The curvature half of the subset filter, `pc_min / pc_max < 0.2` on curvature(k, idx)
     * (include/impl/model.hpp:68-71,96-99), is applied by default with the reference's k = 30
     * (detail::curvature_k, include/impl/scene.hpp:24), like the reference does. disable_curvature_filter()
     * opts out (tangents from another edge detector need not satisfy it). */
    void enable_curvature_filter(uint32_t k = 30) { curvature_k_ = k; }
    void disable_curvature_filter() { curvature_k_ = 0; }

    void init(const sample_parameters& sample_params) { init(subset_t(), sample_params); }

    /* include/impl/model.hpp:15-167. The curvature-ratio half of the subset filter (:96-99) is a "next"
     * row (kNN curvature); pass `curvature_ok` (one flag per cloud point) to apply it. */
    void init(const subset_t& subset, const sample_parameters& sample_params,
              const std::vector<uint8_t>* curvature_ok = nullptr) {
        subset_t sub = subset;
        if (sub.empty()) {
            sub.resize(cloud_->size());
            for (uint32_t i = 0; i < sub.size(); ++i) sub[i] = i;
        }
        subset_.clear();
        if (curvature_k_) { /* the curvature of the points the filter below can reach, in one call */
            std::vector<uint32_t> cand;
            for (uint32_t idx : sub) {
                const Point& p = cloud_->points[idx];
                if (p.getVector3fMap().allFinite() && p.getNormalVector3fMap().allFinite() && tangent(p).allFinite() &&
                    tangent(p).norm() > 0.7f && (!curvature_ok || (*curvature_ok)[idx]))
                    cand.push_back(idx);
            }
            cloud_->prepare_curvature(curvature_k_, cand);
        }
        for (uint32_t idx : sub) {
            const Point& p = cloud_->points[idx];
            bool finite = p.getVector3fMap().allFinite() && p.getNormalVector3fMap().allFinite() &&
                          tangent(p).allFinite();
            if (!finite) continue;
            if (!(tangent(p).norm() > 0.7f)) continue;
            if (curvature_ok && !(*curvature_ok)[idx]) continue;
            if (curvature_k_) {
                curvature_info_t ci = cloud_->curvature(curvature_k_, idx);
                if (!((ci.pc_min / ci.pc_max) < 0.2f)) continue;
            }
            subset_.push_back(idx);
        }
        s_params_ = sample_params;
        /* bounding box, diameter and grid over the caller's subset (:17-38); empty = the whole cloud */
        detail::check(tm_model_set_grid_subset(h_, subset.empty() ? nullptr : subset.data(), (uint32_t)subset.size()),
                      "tm_model_set_grid_subset");
        detail::check(tm_model_build_voxels(h_), "tm_model_build_voxels");
        detail::check(tm_model_build_pairs(h_, subset_.data(), (uint32_t)subset_.size(),
                                           sample_params.min_diameter_factor, sample_params.max_diameter_factor),
                      "tm_model_build_pairs");
        int ext[3];
        detail::check(tm_model_get_grid(h_, ext, to_voxel_.data()), "tm_model_get_grid");
        for (int d = 0; d < 3; ++d) extents_[d] = ext[d];
        detail::check(tm_model_get_feature_bounds(h_, bounds_.lo.data(), bounds_.hi.data()),
                      "tm_model_get_feature_bounds");
        uint64_t entries = 0;
        tm_model_get_pair_counts(h_, &entries, nullptr, nullptr);
        pair_count_ = entries;
        {
            std::lock_guard<std::mutex> lk(table_mutex_);
            table_ready_ = false; /* a second init() rebuilds the mirror on the next query */
        }
        init_ = true;
    }

    /* include/impl/model.hpp:169-178; throws like the reference when called before init(). The reference's
     * workers call query() concurrently (include/impl/scene.hpp:304) and get iterators into the model's own
     * multimap. Here the range points into a host mirror of the device table (8 bytes per entry, canonical
     * order), downloaded once on the first query under a mutex and read-only from then on: query() is
     * re-entrant and a returned range stays valid until the model is re-initialised or destroyed. (The search
     * itself never comes here: it probes the table on the device.) */
    std::pair<pair_iter_t, pair_iter_t> query(const feature_t& f, bool = false) {
        if (!init_) throw std::runtime_error("Cannot query uninitialized model");
        ensure_table();
        const discrete_feature_t key = discretize_feature<Point>(f, bounds_, params_);
        /* keys are sorted lexicographically (tm_model_download_table) */
        size_t lo = 0, hi = t_keys_.size() / 4;
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (std::lexicographical_compare(t_keys_.begin() + 4 * mid, t_keys_.begin() + 4 * mid + 4, key.begin(),
                                             key.end()))
                lo = mid + 1;
            else
                hi = mid;
        }
        const bool hit = lo < t_keys_.size() / 4 && std::equal(key.begin(), key.end(), t_keys_.begin() + 4 * lo);
        const uint32_t* base = t_pairs_.data();
        if (!hit) return {pair_iter_t(base, key), pair_iter_t(base, key)};
        return {pair_iter_t(base + 2 * (size_t)t_offsets_[lo], key), pair_iter_t(base + 2 * (size_t)t_offsets_[lo + 1], key)};
    }

    /* include/impl/model.hpp:180-192 */
    std::optional<voxel_data_t> voxel_query(const vec4f_t& local, bool = false) const {
        uint32_t idx = 0;
        int found = 0;
        detail::check(tm_model_voxel_query(h_, local.v, &idx, &found), "tm_model_voxel_query");
        if (!found) return std::nullopt;
        return voxel_data_t{idx};
    }

    float diameter() const {
        float d = 0.f;
        tm_model_get_info(h_, &d, nullptr, nullptr, nullptr);
        return d;
    }
    uint32_t point_count() const { return (uint32_t)subset_.size(); }
    const vec3i_t& extents() const { return extents_; }
    const mat4f_t& voxel_transform() const { return to_voxel_; }
    int margin() const { return TM_VOXEL_MARGIN; }
    uint64_t pair_count() const { return pair_count_; } /* uninitialised garbage in the reference */
    const typename cloud_t::ConstPtr& cloud() const { return cloud_; }
    const feature_bounds_t& feature_bounds() const { return bounds_; }

    tm_model* handle() const { return h_; }
    tm_ctx* context() const { return ctx_->ctx; }

protected:
    void ensure_table() {
        std::lock_guard<std::mutex> lk(table_mutex_);
        if (table_ready_) return;
        uint64_t entries = 0;
        uint32_t n_keys = 0;
        detail::check(tm_model_get_pair_counts(h_, &entries, nullptr, &n_keys), "tm_model_get_pair_counts");
        t_keys_.assign(4 * (size_t)n_keys, 0u);
        t_offsets_.assign((size_t)n_keys + 1, 0u);
        t_pairs_.assign(2 * (size_t)entries + 2, 0u);
        detail::check(tm_model_download_table(h_, t_keys_.data(), t_offsets_.data(), n_keys, t_pairs_.data(), entries),
                      "tm_model_download_table");
        table_ready_ = true;
    }

    typename cloud_t::ConstPtr cloud_;
    discretization_params params_;
    std::shared_ptr<detail::context_holder> ctx_;
    tm_model* h_ = nullptr;
    bool init_ = false;
    subset_t subset_;
    sample_parameters s_params_;
    vec3i_t extents_;
    mat4f_t to_voxel_;
    feature_bounds_t bounds_;
    uint64_t pair_count_ = 0;
    uint32_t curvature_k_ = 30;
    std::mutex table_mutex_;
    bool table_ready_ = false;
    std::vector<uint32_t> t_keys_, t_offsets_, t_pairs_; /* host mirror of the key table (CSR, canonical order) */
};

}  // namespace triplet_match

#endif
