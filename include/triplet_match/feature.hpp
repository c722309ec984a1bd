/*
 * triplet_match/feature.hpp — pair descriptor of the drop-in API (include/feature:10-36,
 * include/impl/feature.hpp). The curvature arguments are accepted and ignored, as in the reference.
 */
#ifndef TRIPLET_MATCH_FEATURE_HPP_
#define TRIPLET_MATCH_FEATURE_HPP_

#include "discretize.hpp"
#include "pointcloud.hpp"

namespace triplet_match {

static constexpr int feature_dimension = 4; /* include/feature:10 */
typedef std::array<float, feature_dimension> feature_t;
typedef std::array<uint32_t, feature_dimension> discrete_feature_t;

struct feature_bounds_t { /* Eigen::AlignedBox<float,4> as used by the reference */
    feature_t lo, hi;
    feature_bounds_t() {
        lo.fill(std::numeric_limits<float>::max());
        hi.fill(std::numeric_limits<float>::lowest());
    }
    const feature_t& min() const { return lo; }
    const feature_t& max() const { return hi; }
    feature_t& min() { return lo; }
    feature_t& max() { return hi; }
    feature_t diagonal() const {
        feature_t d;
        for (int i = 0; i < feature_dimension; ++i) d[i] = hi[i] - lo[i];
        return d;
    }
    void extend(const feature_t& f) {
        for (int i = 0; i < feature_dimension; ++i) {
            lo[i] = std::min(lo[i], f[i]);
            hi[i] = std::max(hi[i], f[i]);
        }
    }
};

template <typename Point>
using curv_info_t = curvature_info_t;

template <typename Point>
inline std::optional<feature_t> feature(const Point& pnt_0, const Point& pnt_1,
                                        const curv_info_t<Point>& = curvature_info_t(),
                                        const curv_info_t<Point>& = curvature_info_t()) {
    feature_t f;
    tm_feature(pnt_0.raw(), pnt_1.raw(), f.data());
    return f;
}

template <typename Point>
inline discrete_feature_t discretize_feature(const feature_t& f, const feature_bounds_t& bounds,
                                             const discretization_params& params) {
    discrete_feature_t k;
    tm_discretize_feature(f.data(), bounds.lo.data(), bounds.hi.data(), params.distance_step_count,
                          params.angle_step, k.data());
    return k;
}

template <typename Point>
inline bool valid(const feature_t& f, const feature_bounds_t& bounds) {
    return tm_valid(f.data(), bounds.lo.data(), bounds.hi.data()) != 0;
}

inline feature_bounds_t valid_bounds(const feature_bounds_t& bounds, float /*min_angle*/, float /*max_angle*/,
                                     float min_rel_dist, float max_rel_dist) {
    feature_bounds_t nb;
    tm_valid_bounds(bounds.lo.data(), bounds.hi.data(), min_rel_dist, max_rel_dist, nb.lo.data(), nb.hi.data());
    return nb;
}

}  // namespace triplet_match

#endif
