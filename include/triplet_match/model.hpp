/*
 * triplet_match/model.hpp — model<Point> of the drop-in API over the C ABI.
 * Mirrors include/model:15-66 / include/impl/model.hpp: same constructor, init() overloads, query(),
 * voxel_query() and getters; the voxel NN grid, the pair enumeration and the key table live in HBM and
 * are built by HIP kernels (tm_model_build_voxels / tm_model_build_pairs).
 */
#ifndef TRIPLET_MATCH_MODEL_HPP_
#define TRIPLET_MATCH_MODEL_HPP_

#include "feature.hpp"

namespace triplet_match {

constexpr int voxel_multiplicity = 1; /* include/model:13 */

template <typename Point>
class scene;

template <typename Point>
class model {
public:
    typedef std::unique_ptr<model<Point>> uptr_t;
    typedef std::shared_ptr<model<Point>> sptr_t;
    typedef std::shared_ptr<const model<Point>> const_sptr_t;
    typedef pointcloud<Point> cloud_t;
    typedef std::tuple<uint32_t, uint32_t> result_t;
    /* the reference iterates unordered_multimap nodes: `it->second` is the (i, j) tuple */
    typedef std::pair<discrete_feature_t, result_t> entry_t;
    typedef typename std::vector<entry_t>::const_iterator pair_iter_t;
    typedef std::array<uint32_t, voxel_multiplicity> voxel_data_t;

    model(typename cloud_t::ConstPtr cloud, discretization_params params, int device = 0)
        : cloud_(cloud), params_(params), ctx_(detail::default_context(device)) {
        static_assert(sizeof(Point) == 48, "Point must be the 48-byte surfel layout");
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
        init_ = true;
    }

    /* include/impl/model.hpp:169-178; throws like the reference when called before init() */
    std::pair<pair_iter_t, pair_iter_t> query(const feature_t& f, bool = false) {
        if (!init_) throw std::runtime_error("Cannot query uninitialized model");
        uint32_t total = 0;
        detail::check(tm_model_query(h_, f.data(), nullptr, 0, &total), "tm_model_query");
        std::vector<uint32_t> raw(2 * (size_t)total);
        if (total) detail::check(tm_model_query(h_, f.data(), raw.data(), total, &total), "tm_model_query");
        discrete_feature_t key = discretize_feature<Point>(f, bounds_, params_);
        last_query_.resize(total);
        for (uint32_t k = 0; k < total; ++k) last_query_[k] = {key, result_t{raw[2 * k], raw[2 * k + 1]}};
        return {last_query_.cbegin(), last_query_.cend()};
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
    std::vector<entry_t> last_query_;
};

}  // namespace triplet_match

#endif
