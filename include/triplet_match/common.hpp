/*
 * triplet_match/common.hpp — value types of the drop-in C++ API.
 *
 * Mirrors the reference's include/common (typedefs :31-55, tangent() :62-70, sample_parameters :72-82)
 * without Eigen/PCL (neither exists on the target image): small fixed-size vector/matrix structs with
 * the member functions the reference's call sites use, and a Point record that is layout-identical to
 * pcl::PointSurfel (48 bytes). If you build inside a PCL tree you can pass pcl::PointSurfel clouds by
 * reinterpret_cast: the byte layout is the same.
 */
#ifndef TRIPLET_MATCH_COMMON_HPP_
#define TRIPLET_MATCH_COMMON_HPP_

#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#if defined(__GLIBC__)
#include <malloc.h>
#endif

#include "../tm_hip.h"

namespace triplet_match {

constexpr bool scale_invariant = false; /* include/common:29 */

struct vec3f_t {
    float v[3] = {0.f, 0.f, 0.f};
    vec3f_t() = default;
    vec3f_t(float x, float y, float z) : v{x, y, z} {}
    float& operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
    float* data() { return v; }
    const float* data() const { return v; }
    float dot(const vec3f_t& o) const { return (v[0] * o.v[0] + v[1] * o.v[1]) + v[2] * o.v[2]; }
    float squaredNorm() const { return dot(*this); }
    float norm() const { return std::sqrt(squaredNorm()); }
    vec3f_t cross(const vec3f_t& o) const {
        return {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]};
    }
    vec3f_t normalized() const {
        float q = squaredNorm();
        if (!(q > 0.f)) return *this;
        float n = std::sqrt(q);
        return {v[0] / n, v[1] / n, v[2] / n};
    }
    void normalize() { *this = normalized(); }
    vec3f_t operator+(const vec3f_t& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
    vec3f_t operator-(const vec3f_t& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
    vec3f_t operator*(float s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
    bool allFinite() const { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }
};

struct vec4f_t {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    vec4f_t() = default;
    vec4f_t(float x, float y, float z, float w) : v{x, y, z, w} {}
    float& operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
    vec3f_t head3() const { return {v[0], v[1], v[2]}; }
};
inline vec4f_t homogeneous(const vec3f_t& p) { return {p[0], p[1], p[2], 1.f}; }

struct vec3i_t {
    int v[3] = {0, 0, 0};
    int& operator[](int i) { return v[i]; }
    int operator[](int i) const { return v[i]; }
};

/* column-major 4x4 like Eigen::Matrix4f */
struct mat4f_t {
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    static mat4f_t Identity() { return mat4f_t(); }
    float& operator()(int r, int c) { return m[c * 4 + r]; }
    float operator()(int r, int c) const { return m[c * 4 + r]; }
    float* data() { return m; }
    const float* data() const { return m; }
    vec4f_t operator*(const vec4f_t& p) const {
        vec4f_t o;
        for (int r = 0; r < 4; ++r)
            o[r] = ((m[r] * p[0] + m[4 + r] * p[1]) + m[8 + r] * p[2]) + m[12 + r] * p[3];
        return o;
    }
    mat4f_t operator*(const mat4f_t& b) const {
        mat4f_t o;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
                for (int k = 0; k < 4; ++k) s += (*this)(r, k) * b(k, c);
                o(r, c) = s;
            }
        return o;
    }
    /* general inverse (Gauss-Jordan in double); the reference uses Eigen's .inverse() at
     * include/impl/scene.hpp:92 to return model->scene transforms */
    mat4f_t inverse() const {
        double a[4][8];
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                a[r][c] = (*this)(r, c);
                a[r][4 + c] = r == c ? 1.0 : 0.0;
            }
        for (int c = 0; c < 4; ++c) {
            int piv = c;
            for (int r = c + 1; r < 4; ++r)
                if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
            for (int k = 0; k < 8; ++k) std::swap(a[c][k], a[piv][k]);
            double d = a[c][c];
            for (int k = 0; k < 8; ++k) a[c][k] /= d;
            for (int r = 0; r < 4; ++r) {
                if (r == c) continue;
                double f = a[r][c];
                for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k];
            }
        }
        mat4f_t o;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) o(r, c) = (float)a[r][4 + c];
        return o;
    }
};

typedef std::vector<uint32_t> subset_t; /* include/common:55 */

/* Layout of pcl::PointSurfel (48 bytes): data[4], data_n[4], data_c[4] = {rgba, radius, confidence,
 * curvature}. The reference keeps the crease tangent in data_c[1..3] (include/common:62-70). */
struct alignas(16) PointSurfel {
    float x = 0.f, y = 0.f, z = 0.f, w = 1.f;
    float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, normal_w = 0.f;
    union {
        struct {
            uint32_t rgba;
            float radius;
            float confidence;
            float curvature;
        };
        float data_c[4];
    };
    PointSurfel() : data_c{0.f, 0.f, 0.f, 0.f} {}
    vec3f_t getVector3fMap() const { return {x, y, z}; }
    vec3f_t getNormalVector3fMap() const { return {normal_x, normal_y, normal_z}; }
    const float* raw() const { return &x; }
    float* raw() { return &x; }
};
static_assert(sizeof(PointSurfel) == 48, "PointSurfel must be the 48-byte PCL layout");

inline vec3f_t tangent(const PointSurfel& p) { return {p.data_c[1], p.data_c[2], p.data_c[3]}; }
inline void set_tangent(PointSurfel& p, const vec3f_t& t) {
    p.data_c[1] = t[0];
    p.data_c[2] = t[1];
    p.data_c[3] = t[2];
}

struct sample_parameters { /* include/common:72-82; only the diameter factors and force_up are live */
    float min_triplet_ratio = 0.f;
    float max_triplet_ratio = 0.f;
    float search_min_scale = 0.f;
    float search_max_scale = 0.f;
    float min_diameter_factor = 0.2f;
    float max_diameter_factor = 1.0f;
    float min_orthogonality = 0.f;
    float redundancy_factor = 0.f;
    bool force_up = false;
};

namespace detail {
inline void check(int rc, const char* what) {
    if (rc != TM_OK) throw std::runtime_error(std::string(what) + ": " + tm_last_error());
}
/* one HIP context per device, shared by the models and scenes of this process */
struct context_holder {
    tm_ctx* ctx = nullptr;
    explicit context_holder(int device) { check(tm_ctx_create(device, &ctx), "tm_ctx_create"); }
    ~context_holder() { tm_ctx_destroy(ctx); }
};
/* A search allocates and frees vectors of 0.4-4 MB per call (correspondence lists of the candidates, masks). glibc
 * serves blocks above M_MMAP_THRESHOLD (128 KB until the process happens to free a larger mapping, then up to 32 MB)
 * by mmap/munmap: every such vector is then mapped, page-faulted and unmapped again — 2-3 ms of a 10-ms search, and
 * whether a process pays it depended on what it had freed before. The threshold is therefore pinned once (glibc only;
 * TM_KEEP_MALLOC_DEFAULTS in the environment leaves the allocator alone). */
inline void tune_allocator_once() {
#if defined(__GLIBC__)
    static const bool done = [] {
        if (std::getenv("TM_KEEP_MALLOC_DEFAULTS") == nullptr) {
            mallopt(M_MMAP_THRESHOLD, 32 << 20);
            mallopt(M_TRIM_THRESHOLD, 128 << 20);
        }
        return true;
    }();
    (void)done;
#endif
}
inline std::shared_ptr<context_holder> default_context(int device = 0) {
    tune_allocator_once();
    static std::mutex guard; /* models and scenes may be constructed on several threads at once */
    std::lock_guard<std::mutex> lk(guard);
    static std::weak_ptr<context_holder> cached[16];
    auto sp = cached[device & 15].lock();
    if (!sp) {
        sp = std::make_shared<context_holder>(device);
        cached[device & 15] = sp;
    }
    return sp;
}
}  // namespace detail

/* A context of its own (one HIP stream, one set of work buffers, one lock) for callers that run several searches at
 * once: model<Point> and scene<Point> constructed WITHOUT one share the process-wide default context of their device,
 * whose calls serialise; objects constructed on different contexts run side by side on the GPU, as the reference's
 * objects do on the CPU (no shared state between two scene objects, include/impl/scene.hpp:569-575). A scene is searched
 * against a model of the SAME context. */
typedef std::shared_ptr<detail::context_holder> context_t;
inline context_t make_context(int device = 0) {
    detail::tune_allocator_once();
    return std::make_shared<detail::context_holder>(device);
}

}  // namespace triplet_match

#endif
