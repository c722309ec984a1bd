/*
 * triplet_match/traits.hpp — the *_traits plugin surface (identity / plane / plane2 / cylinder) of the
 * drop-in API. Mirrors include/{identity,plane,plane2,cylinder}_traits (+ the impl/ headers): state_t,
 * handle_t, const_handle_t, sample_count, samples_t and the seven statics, all over the C ABI
 * (tm_traits_*). In this snapshot of the reference the traits are compiled but not wired into
 * model<>/scene<>; here a projected search is wired by projecting the clouds with
 * tm_traits_project_cloud and running the same kernels in uvw space (DESIGN.md "Traits").
 */
#ifndef TRIPLET_MATCH_TRAITS_HPP_
#define TRIPLET_MATCH_TRAITS_HPP_

#include <limits>
#include <random>
#include <stdexcept>
#include <cmath>
#include <tuple>
#include <utility>

#include "pointcloud.hpp"

namespace triplet_match {

/* repeat_t<T, N>: std::tuple of N T's (include/cartesian_tuple:29-51) */
namespace detail {
template <typename T, size_t>
using depend_t = T;
template <typename T, size_t N, typename I = std::make_index_sequence<N>>
struct repeat;
template <typename T, size_t N, size_t... Is>
struct repeat<T, N, std::index_sequence<Is...>> {
    using type = std::tuple<depend_t<T, Is>...>;
};
}  // namespace detail
template <typename T, size_t N>
using repeat_t = typename detail::repeat<T, N>::type;

struct bbox3_t {
    vec3f_t lo, hi;
};

namespace detail {

struct projector_state {
    float radius = 0.f;
    float threshold = 0.f;
    vec3f_t origin;
    mat4f_t g2l;
    mat4f_t l2g;
    bbox3_t uvw_bounds;
    void pack(float* st) const {
        st[0] = radius;
        st[1] = threshold;
        for (int d = 0; d < 3; ++d) st[2 + d] = origin[d];
        memcpy(st + 5, g2l.m, 64);
        memcpy(st + 21, l2g.m, 64);
    }
    void unpack(const float* st) {
        radius = st[0];
        threshold = st[1];
        origin = {st[2], st[3], st[4]};
        memcpy(g2l.m, st + 5, 64);
        memcpy(l2g.m, st + 21, 64);
    }
};

/* principal frame of a point set: rows of g2l = eigenvectors of the scatter matrix by descending
 * eigenvalue (the right singular vectors the reference takes from Eigen::JacobiSVD,
 * include/impl/plane_traits.hpp:12-31; their signs are not pinned by the reference) */
template <typename Point>
inline void principal_frame(const pointcloud<Point>& cloud, const subset_t& subset, projector_state& st) {
    std::vector<uint32_t> idx(subset.begin(), subset.end());
    if (idx.empty()) {
        idx.resize(cloud.size());
        for (uint32_t i = 0; i < idx.size(); ++i) idx[i] = i;
    }
    double c[3] = {0, 0, 0};
    for (uint32_t i : idx) {
        c[0] += cloud.points[i].x;
        c[1] += cloud.points[i].y;
        c[2] += cloud.points[i].z;
    }
    for (double& v : c) v /= (double)idx.size();
    double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (uint32_t i : idx) {
        double d[3] = {cloud.points[i].x - c[0], cloud.points[i].y - c[1], cloud.points[i].z - c[2]};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) S[a][b] += d[a] * d[b];
    }
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep)
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (std::fabs(S[p][q]) < 1e-300) continue;
                double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) {
                    double a = S[k][p], b = S[k][q];
                    S[k][p] = cs * a - sn * b;
                    S[k][q] = sn * a + cs * b;
                }
                for (int k = 0; k < 3; ++k) {
                    double a = S[p][k], b = S[q][k];
                    S[p][k] = cs * a - sn * b;
                    S[q][k] = sn * a + cs * b;
                }
                for (int k = 0; k < 3; ++k) {
                    double a = V[k][p], b = V[k][q];
                    V[k][p] = cs * a - sn * b;
                    V[k][q] = sn * a + cs * b;
                }
            }
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int a, int b) { return S[a][a] > S[b][b]; });
    st.origin = {(float)c[0], (float)c[1], (float)c[2]};
    st.g2l = mat4f_t::Identity();
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) st.g2l(r, k) = (float)V[k][order[r]];
    for (int r = 0; r < 3; ++r)
        st.g2l(r, 3) = -((st.g2l(r, 0) * st.origin[0] + st.g2l(r, 1) * st.origin[1]) + st.g2l(r, 2) * st.origin[2]);
    st.l2g = st.g2l.inverse();
    st.threshold = 0.f;
    for (uint32_t i : idx) {
        const auto& p = cloud.points[i];
        float w = ((st.g2l(2, 0) * p.x + st.g2l(2, 1) * p.y) + st.g2l(2, 2) * p.z) + st.g2l(2, 3);
        st.threshold = std::max(st.threshold, std::fabs(w));
    }
    st.threshold *= 2.f; /* include/impl/plane_traits.hpp:34-39 */
}

template <typename Point, int Kind, int SampleCount>
struct traits_impl {
    typedef pointcloud<Point> cloud_t;
    typedef projector_state state_t;
    typedef std::shared_ptr<state_t> handle_t;
    typedef std::shared_ptr<const state_t> const_handle_t;
    static constexpr int sample_count = SampleCount;
    typedef repeat_t<Point, SampleCount> samples_t;
    static constexpr int kind = Kind;

    static handle_t init_from_samples(const const_handle_t& model_handle, const samples_t& samples) {
        float ms[TM_STATE_FLOATS] = {0}, st[TM_STATE_FLOATS] = {0};
        if (model_handle) model_handle->pack(ms);
        float flat[12 * SampleCount];
        flatten(samples, flat, std::make_index_sequence<SampleCount>());
        if (tm_traits_init_from_samples(Kind, ms, flat, st)) return nullptr; /* rejected sample set */
        auto h = std::make_shared<state_t>();
        h->unpack(st);
        return h;
    }
    static std::optional<vec3f_t> project(const const_handle_t& handle, const vec3f_t& xyz) {
        float st[TM_STATE_FLOATS] = {0};
        if (handle) handle->pack(st);
        vec3f_t uvw;
        if (!tm_traits_project(Kind, st, xyz.v, uvw.v)) return std::nullopt;
        return uvw;
    }
    static vec3f_t unproject(const const_handle_t& handle, const vec3f_t& uvw) {
        float st[TM_STATE_FLOATS] = {0};
        if (handle) handle->pack(st);
        vec3f_t xyz;
        tm_traits_unproject(Kind, st, uvw.v, xyz.v);
        return xyz;
    }
    static vec3f_t tangent(const const_handle_t& handle, const Point& pnt) {
        float st[TM_STATE_FLOATS] = {0};
        if (handle) handle->pack(st);
        vec3f_t o;
        tm_traits_tangent(Kind, st, pnt.raw(), o.v);
        return o;
    }
    static vec3f_t normal(const const_handle_t& handle, const Point& pnt) {
        float st[TM_STATE_FLOATS] = {0};
        if (handle) handle->pack(st);
        vec3f_t o;
        tm_traits_normal(Kind, st, pnt.raw(), o.v);
        return o;
    }
    static float intrinsic_distance(const const_handle_t& handle, const vec3f_t& p0, const vec3f_t& p1) {
        float st[TM_STATE_FLOATS] = {0};
        if (handle) handle->pack(st);
        return tm_traits_intrinsic_distance(Kind, st, p0.v, p1.v);
    }

private:
    template <size_t... Is>
    static void flatten(const samples_t& s, float* out, std::index_sequence<Is...>) {
        (void)std::initializer_list<int>{(memcpy(out + 12 * Is, std::get<Is>(s).raw(), 48), 0)...};
    }
};

}  // namespace detail

/* include/identity_traits:11-41 */
template <typename Point>
struct identity_traits : detail::traits_impl<Point, TM_TRAITS_IDENTITY, 1> {
    typedef detail::traits_impl<Point, TM_TRAITS_IDENTITY, 1> base;
    static typename base::handle_t init_from_model(typename base::cloud_t::ConstPtr, const subset_t& = subset_t()) {
        return std::make_shared<typename base::state_t>();
    }
};

/* include/plane_traits, impl/plane_traits.hpp:5-41 */
template <typename Point>
struct plane_traits : detail::traits_impl<Point, TM_TRAITS_PLANE, 1> {
    typedef detail::traits_impl<Point, TM_TRAITS_PLANE, 1> base;
    static typename base::handle_t init_from_model(typename base::cloud_t::ConstPtr cloud,
                                                   const subset_t& subset = subset_t()) {
        auto h = std::make_shared<typename base::state_t>();
        detail::principal_frame(*cloud, subset, *h);
        return h;
    }
};

/* include/plane2_traits, impl/plane2_traits.hpp:5-45 */
template <typename Point>
struct plane2_traits : detail::traits_impl<Point, TM_TRAITS_PLANE2, 3> {
    typedef detail::traits_impl<Point, TM_TRAITS_PLANE2, 3> base;
    static typename base::handle_t init_from_model(typename base::cloud_t::ConstPtr cloud,
                                                   const subset_t& subset = subset_t()) {
        auto h = std::make_shared<typename base::state_t>();
        detail::principal_frame(*cloud, subset, *h);
        return h;
    }
};

/* include/cylinder_traits. init_from_model in the reference hands the model cloud to PCL's MSAC cylinder fit
 * (pcl::MEstimatorSampleConsensus over pcl::SampleConsensusModelCylinder, impl/cylinder_traits.hpp:10-26: threshold
 * 5 x resolution, probability 0.99, no refinement) and builds the frame from the winning coefficients (:28-39). PCL
 * is not in the reference tree, so its published algorithm is restated here: candidate = the cylinder through two
 * sampled points with their normals (the closest approach of the two normal lines — the same construction
 * init_from_samples uses, :53-97, which the reference took from PCL's computeModelCoefficients); distance of a point
 * = | distance to the axis - radius | (normal weight 0, PCL's default); penalty = sum of min(distance, threshold);
 * the candidate with the lowest penalty wins; after every improvement the iteration bound becomes
 * log(1 - 0.99) / log(1 - w^2) with w = its inlier fraction (at most 10,000, MSAC's max_iterations_). What cannot be
 * restated is PCL's sample sequence (boost RNG, unpinned version): samples are drawn with mt19937 seed 13, like every
 * other draw of this library — the cylinder found is one PCL could have found, not necessarily the one it would. */
template <typename Point>
struct cylinder_traits : detail::traits_impl<Point, TM_TRAITS_CYLINDER, 2> {
    typedef detail::traits_impl<Point, TM_TRAITS_CYLINDER, 2> base;
    static typename base::handle_t init_from_model(typename base::cloud_t::ConstPtr cloud,
                                                   const subset_t& subset = subset_t(), uint32_t seed = 13) {
        if (!cloud || cloud->size() < 2) throw std::runtime_error("cylinder_traits::init_from_model: fewer than two points");
        std::vector<uint32_t> indices(subset.begin(), subset.end());
        if (indices.empty()) { /* :13-16 */
            indices.resize(cloud->size());
            for (uint32_t i = 0; i < (uint32_t)indices.size(); ++i) indices[i] = i;
        }
        const float threshold = 5.f * cloud->resolution(); /* :19 */
        const size_t n = indices.size();
        if (n < 2) throw std::runtime_error("cylinder_traits::init_from_model: fewer than two points in the subset");
        float ms[TM_STATE_FLOATS] = {0};
        ms[1] = threshold;
        std::mt19937 rng(seed);
        double best_penalty = std::numeric_limits<double>::max();
        float best[TM_STATE_FLOATS] = {0};
        bool have = false;
        double k = 1.0;
        const uint32_t max_iterations = 10000;
        uint32_t iterations = 0, skipped = 0;
        while ((double)iterations < k && iterations < max_iterations && skipped < 10u * max_iterations) {
            const uint32_t a = indices[rng() % n], b = indices[rng() % n];
            if (a == b) { /* PCL's isSampleGood: two distinct points */
                ++skipped;
                continue;
            }
            float flat[24], st[TM_STATE_FLOATS] = {0};
            memcpy(flat, cloud->points[a].raw(), 48);
            memcpy(flat + 12, cloud->points[b].raw(), 48);
            if (tm_traits_init_from_samples(TM_TRAITS_CYLINDER, ms, flat, st) || !(st[0] > 0.f) || !std::isfinite(st[0])) {
                ++skipped; /* degenerate sample (parallel normals through one line, non-finite point) */
                continue;
            }
            /* axis = third row of g2l (column-major 4x4 at st + 5), a point on it = origin */
            const vec3f_t o(st[2], st[3], st[4]), ax(st[5 + 2], st[5 + 6], st[5 + 10]);
            double penalty = 0.0;
            size_t inliers = 0;
            for (uint32_t i : indices) {
                const Point& p = cloud->points[i];
                const vec3f_t d(p.x - o[0], p.y - o[1], p.z - o[2]);
                const float dist = std::fabs(ax.cross(d).norm() - st[0]);
                if (dist < threshold) { /* NaN fails: counted at the threshold like any outlier */
                    penalty += (double)dist;
                    ++inliers;
                } else {
                    penalty += (double)threshold;
                }
            }
            if (penalty < best_penalty) {
                best_penalty = penalty;
                memcpy(best, st, sizeof best);
                have = true;
                /* the iteration bound from the inlier fraction (sample size 2), clamped as PCL clamps it */
                const double w = (double)inliers / (double)n;
                double p_no_outliers = 1.0 - w * w;
                p_no_outliers = std::max(std::numeric_limits<double>::epsilon(), p_no_outliers);
                p_no_outliers = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no_outliers);
                k = std::log(1.0 - 0.99) / std::log(p_no_outliers);
            }
            ++iterations;
        }
        if (!have) throw std::runtime_error("cylinder_traits::init_from_model: no cylinder through any sampled pair");
        /* :28-39: frame from the axis via unitOrthogonal, as init_from_axis builds it */
        return init_from_axis(vec3f_t(best[2], best[3], best[4]), vec3f_t(best[5 + 2], best[5 + 6], best[5 + 10]), best[0],
                              threshold);
    }
    /* threshold = 5 * resolution in the reference (:19) */
    static typename base::handle_t init_from_axis(const vec3f_t& origin, const vec3f_t& axis_dir, float radius,
                                                  float threshold) {
        auto h = std::make_shared<typename base::state_t>();
        vec3f_t axis = axis_dir.normalized();
        /* Eigen unitOrthogonal(), :29 */
        vec3f_t ref;
        auto small = [](float a, float b) { return std::fabs(a) <= std::fabs(b) * 1e-5f; };
        if (!small(axis[0], axis[2]) || !small(axis[1], axis[2])) {
            float inv = 1.f / std::sqrt(axis[0] * axis[0] + axis[1] * axis[1]);
            ref = {-axis[1] * inv, axis[0] * inv, 0.f};
        } else {
            float inv = 1.f / std::sqrt(axis[1] * axis[1] + axis[2] * axis[2]);
            ref = {0.f, -axis[2] * inv, axis[1] * inv};
        }
        vec3f_t co = ref.cross(axis);
        h->origin = origin;
        h->radius = radius;
        h->threshold = threshold;
        h->g2l = mat4f_t::Identity();
        for (int k = 0; k < 3; ++k) {
            h->g2l(0, k) = ref[k];
            h->g2l(1, k) = co[k];
            h->g2l(2, k) = axis[k];
        }
        for (int r = 0; r < 3; ++r)
            h->g2l(r, 3) = -((h->g2l(r, 0) * origin[0] + h->g2l(r, 1) * origin[1]) + h->g2l(r, 2) * origin[2]);
        h->l2g = h->g2l.inverse();
        return h;
    }
};

}  // namespace triplet_match

#endif
