/*
 * triplet_match/pointcloud.hpp — cloud container of the drop-in API.
 *
 * Mirrors the parts of the reference's pointcloud<Point> (include/pointcloud:13-93) the hot path
 * consumes: points, size(), resolution() (mean 1-NN distance, include/impl/pointcloud.hpp:66-82),
 * diameter() (:84-97). The kd-tree queries, curvature() and the range-v3 view adaptors are "next" rows
 *  and are not provided; resolution() uses an exact uniform-grid nearest-neighbour scan
 * on the host (a one-off preprocessing input, not the hot path) or a caller-supplied value.
 */
#ifndef TRIPLET_MATCH_POINTCLOUD_HPP_
#define TRIPLET_MATCH_POINTCLOUD_HPP_

#include <algorithm>
#include <limits>
#include <unordered_map>

#include "common.hpp"

namespace triplet_match {

template <typename Point>
class pointcloud {
public:
    typedef std::shared_ptr<pointcloud<Point>> Ptr;
    typedef std::shared_ptr<const pointcloud<Point>> ConstPtr;

    std::vector<Point> points;

    static Ptr empty() { return std::make_shared<pointcloud<Point>>(); }

    size_t size() const { return points.size(); }
    void push_back(const Point& p) {
        points.push_back(p);
        resolution_ = diameter_ = -1.f;
    }
    const Point& operator[](size_t i) const { return points[i]; }
    Point& operator[](size_t i) { return points[i]; }
    typename std::vector<Point>::const_iterator begin() const { return points.begin(); }
    typename std::vector<Point>::const_iterator end() const { return points.end(); }

    void set_resolution(float r) { resolution_ = r; }

    /* mean distance to the nearest other point; running mean accum + (val - accum)/(++i) in float as
     * the reference's average() does (include/common:104-115) */
    float resolution() const {
        if (resolution_ >= 0.f) return resolution_;
        const size_t n = points.size();
        if (n < 2) return resolution_ = 0.f;
        float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
                       std::numeric_limits<float>::max()};
        float mx[3] = {-mn[0], -mn[0], -mn[0]};
        for (const auto& p : points) {
            const float c[3] = {p.x, p.y, p.z};
            if (!(std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]))) continue;
            for (int d = 0; d < 3; ++d) {
                mn[d] = std::min(mn[d], c[d]);
                mx[d] = std::max(mx[d], c[d]);
            }
        }
        double vol_side = std::max({(double)mx[0] - mn[0], (double)mx[1] - mn[1], (double)mx[2] - mn[2], 1e-12});
        /* surface-like clouds: ~ sqrt(n) points per side */
        double cell = vol_side / std::max(1.0, std::sqrt((double)n) / 2.0);
        auto key = [&](const Point& p, int d) { return (int64_t)std::floor(((&p.x)[d] - mn[d]) / cell); };
        auto pack = [](int64_t a, int64_t b, int64_t c) { return (uint64_t)((a + 1) << 42) ^ (uint64_t)((b + 1) << 21) ^ (uint64_t)(c + 1); };
        std::unordered_map<uint64_t, std::vector<uint32_t>> grid;
        grid.reserve(n);
        for (uint32_t i = 0; i < n; ++i) {
            const Point& p = points[i];
            if (!(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) continue;
            grid[pack(key(p, 0), key(p, 1), key(p, 2))].push_back(i);
        }
        float accum = 0.f;
        uint32_t count = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const Point& p = points[i];
            if (!(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) continue;
            const int64_t k0 = key(p, 0), k1 = key(p, 1), k2 = key(p, 2);
            float best = std::numeric_limits<float>::max();
            /* shells of cells at Chebyshev distance `ring`; after shell r every unvisited point is
             * farther than r * cell, so the scan is exact once sqrt(best) <= r * cell */
            for (int ring = 0; ring <= 4096; ++ring) {
                for (int64_t a = k0 - ring; a <= k0 + ring; ++a)
                    for (int64_t b = k1 - ring; b <= k1 + ring; ++b)
                        for (int64_t c = k2 - ring; c <= k2 + ring; ++c) {
                            if (std::max({std::llabs(a - k0), std::llabs(b - k1), std::llabs(c - k2)}) != ring) continue;
                            auto it = grid.find(pack(a, b, c));
                            if (it == grid.end()) continue;
                            for (uint32_t j : it->second) {
                                if (j == i) continue;
                                float dx = points[j].x - p.x, dy = points[j].y - p.y, dz = points[j].z - p.z;
                                best = std::min(best, (dx * dx + dy * dy) + dz * dz);
                            }
                        }
                if (best < std::numeric_limits<float>::max() && std::sqrt((double)best) <= (double)ring * cell) break;
            }
            if (best == std::numeric_limits<float>::max()) continue;
            accum = accum + (std::sqrt(best) - accum) / (float)(++count);
        }
        return resolution_ = accum;
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
};

}  // namespace triplet_match

#endif
