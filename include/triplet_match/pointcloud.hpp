/*
 * triplet_match/pointcloud.hpp — cloud container of the drop-in API.
 *
 * Mirrors the parts of the reference's pointcloud<Point> (include/pointcloud:13-93) the hot path
 * consumes: points, size(), from_pcd() (include/impl/pointcloud.hpp:58-64), resolution() (mean 1-NN
 * distance, :66-82; tm_cloud_resolution, GPU kd-tree 2-NN, unless set_resolution() supplied a value),
 * diameter() (:84-97) and curvature(k, idx) (:200-204; tm_cloud_curvature). The generic knn_* and
 * radius_search_* accessors and the range-v3 view adaptors are not on the path and are not provided
 * (the search's radius neighbourhood lives in scene.hpp).
 */
#ifndef TRIPLET_MATCH_POINTCLOUD_HPP_
#define TRIPLET_MATCH_POINTCLOUD_HPP_

#include <algorithm>
#include <limits>

#include "common.hpp"
#include "pcd_io.hpp"

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

    size_t size() const { return points.size(); }
    void push_back(const Point& p) {
        points.push_back(p);
        resolution_ = diameter_ = -1.f;
        curv_k_ = 0;
    }
    const Point& operator[](size_t i) const { return points[i]; }
    Point& operator[](size_t i) { return points[i]; }
    typename std::vector<Point>::const_iterator begin() const { return points.begin(); }
    typename std::vector<Point>::const_iterator end() const { return points.end(); }

    void set_resolution(float r) { resolution_ = r; }

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
     * normals (tm_cloud_curvature, GPU); all points are computed on first use and cached per k */
    curvature_info_t curvature(uint32_t k, uint32_t idx) const {
        if (curv_k_ != k || curv_.size() != 5 * points.size()) {
            curv_.assign(5 * points.size(), 0.f);
            auto ctx = detail::default_context(0);
            if (!points.empty())
                detail::check(tm_cloud_curvature(ctx->ctx, points[0].raw(), (uint32_t)points.size(), k, curv_.data()),
                              "tm_cloud_curvature");
            curv_k_ = k;
        }
        curvature_info_t c;
        c.pc_max_dir = {curv_[5 * (size_t)idx], curv_[5 * (size_t)idx + 1], curv_[5 * (size_t)idx + 2]};
        c.pc_min = curv_[5 * (size_t)idx + 3];
        c.pc_max = curv_[5 * (size_t)idx + 4];
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
    mutable float resolution_ = -1.f;
    mutable float diameter_ = -1.f;
    mutable std::vector<float> curv_;
    mutable uint32_t curv_k_ = 0;
};

}  // namespace triplet_match

#endif
