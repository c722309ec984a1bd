/*
 * tm_oracle.cpp — CPU ORACLE: plain C++17 restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see tm_oracle.h). PARITY UNPINNED: the reference has no tests or golden
 * vectors and is unbuildable here (Eigen3/PCL/FLANN/Boost/range-v3 absent); this file restates the
 * algorithm text. Citations "ref:" are relative to /root/reference.
 *
 * Arithmetic contract: everything fp32 except the score sum (fp64);
 * no FMA contraction (build with -ffp-contract=off, no -march); fixed association:
 *   dot(a,b)      = (a0*b0 + a1*b1) + a2*b2
 *   |a|^2         = (a0*a0 + a1*a1) + a2*a2          norm = sqrtf(|a|^2)
 *   mat4*vec4 row = ((m0*x + m4*y) + m8*z) + m12*w   (column-major, cf. ref: opencl/util.cl:4-7)
 *   mat3*vec3 row = (r0*x + r1*y) + r2*z
 * atan2f is NOT libm's: the oracle and the HIP kernels each implement the same published
 * Cephes-style atanf reduction (tmo_atan2f below) so that angle bins agree bit for bit.
 */
#include "tm_oracle.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct v3 {
    float x, y, z;
};
inline v3 sub(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline v3 add(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline v3 mul(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline v3 divs(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float sqn(v3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
inline float norm(v3 a) { return sqrtf(sqn(a)); }
inline v3 cross(v3 a, v3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
/* Eigen normalize(): z = squaredNorm(); if (z > 0) v /= sqrt(z) */
inline v3 normalized(v3 a) {
    float z = sqn(a);
    if (z > 0.f) return divs(a, sqrtf(z));
    return a;
}

inline v3 s_pos(const float* s) { return {s[0], s[1], s[2]}; }
inline v3 s_nrm(const float* s) { return {s[4], s[5], s[6]}; }
inline v3 s_tgt(const float* s) { return {s[9], s[10], s[11]}; } /* ref: include/common:62-70 */

/* column-major 4x4 * (x,y,z,w) */
inline void mat4_mul(const float* m, float x, float y, float z, float w, float out[4]) {
    for (int r = 0; r < 4; ++r) out[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}
/* top-left 3x3 of a column-major 4x4 times vec3 */
inline v3 mat3_mul(const float* m, v3 v) {
    return {(m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
            (m[2] * v.x + m[6] * v.y) + m[10] * v.z};
}

constexpr float kPiF = 3.14159265358979323846f;
constexpr float kPiO2F = 1.57079632679489661923f;
constexpr float kPiO4F = 0.78539816339744830962f;

/* Cephes atanf range reduction + degree-4 odd polynomial (public domain algorithm, S. Moshier). */
inline float atanf_def(float xx) {
    float sign = 1.f;
    float x = xx;
    if (xx < 0.f) {
        sign = -1.f;
        x = -xx;
    }
    float y;
    if (x > 2.414213562373095f) {
        y = kPiO2F;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = kPiO4F;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.f;
    }
    float z = x * x;
    float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z -
               3.33329491539e-1f) *
                  z * x +
              x;
    y = y + p;
    return sign * y;
}

}  // namespace

extern "C" {

/* ---------------------------------------------------------------------------------------------
 * a4: MurmurHash3_x86_32 over u32 words. ref: include/impl/discretize.hpp:5-45 (murmur<Dim>,
 * murmur_seed = 42 at :10, length xor 4*Dim at :33, fmix32 at :35-39).
 * ------------------------------------------------------------------------------------------- */
uint32_t tmo_murmur3_32(const uint32_t* words, int n_words, uint32_t seed) {
    uint32_t h1 = seed;
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    for (int i = 0; i < n_words; ++i) {
        uint32_t k1 = words[i];
        k1 *= c1;
        k1 = (k1 << 15) | (k1 >> 17);
        k1 *= c2;
        h1 ^= k1;
        h1 = (h1 << 13) | (h1 >> 19);
        h1 = h1 * 5u + 0xe6546b64u;
    }
    h1 ^= (uint32_t)(4 * n_words);
    h1 ^= h1 >> 16;
    h1 *= 0x85ebca6bu;
    h1 ^= h1 >> 13;
    h1 *= 0xc2b2ae35u;
    h1 ^= h1 >> 16;
    return h1;
}
uint32_t tmo_murmur_key(const uint32_t* key, int dim) { return tmo_murmur3_32(key, dim, 42u); }

/* a3: ref: src/discretize.cpp:20-26. NaN nval (range 0) falls through both tests; the x86-64
 * float->u32 cast of NaN yields 0, which this restatement defines explicitly. */
uint32_t tmo_discretize_range(float value, float min_value, float range_value, uint32_t steps) {
    float nval = (value - min_value) / range_value;
    if (nval < 0.f) return 0;
    if (nval >= 1.f) return steps - 1;
    float scaled = nval * (float)steps;
    if (!(scaled == scaled)) return 0;
    return (uint32_t)scaled;
}
/* a3: ref: src/discretize.cpp:28-30 */
uint32_t tmo_discretize_step(float value, float step_size) {
    float q = value / step_size;
    if (!(q == q) || q < 0.f) return 0;
    return (uint32_t)q;
}

/* atan2f as defined for this project (see file header). x == 0 returns exactly +-pi/2 like libm. */
float tmo_atan2f(float y, float x) {
    if (x == 0.f) {
        if (y > 0.f) return kPiO2F;
        if (y < 0.f) return -kPiO2F;
        return 0.f;
    }
    float w = 0.f;
    if (x < 0.f) w = (y < 0.f) ? -kPiF : kPiF;
    float z = atanf_def(y / x);
    return w + z;
}

/* a2: ref: include/impl/feature.hpp:5-8 (angle), :16-34 (feature). Curvature args are ignored by
 * the reference; the optional is always engaged. */
void tmo_feature(const float* s0, const float* s1, float f[4]) {
    v3 p0 = s_pos(s0), p1 = s_pos(s1);
    v3 t0 = s_tgt(s0), t1 = s_tgt(s1);
    v3 d0 = sub(p1, p0);
    f[0] = norm(d0);
    f[1] = tmo_atan2f(norm(cross(d0, t0)), fabsf(dot(d0, t0)));
    f[2] = tmo_atan2f(norm(cross(d0, t1)), fabsf(dot(d0, t1)));
    f[3] = f[0];
}

/* The one place where the restatement (and the HIP kernels) knowingly compute something else than the reference's
 * BINARY: the reference calls glibc's atan2f (include/impl/feature.hpp:7), we call tmo_atan2f (<= 1.1 ulp apart).
 * This census says how often that changes a result: over the ordered pairs (i, j) of `subset` that pass the model's
 * pair filters (include/impl/model.hpp:104-112; ij == NULL), or over an explicit pair list, the two descriptor angles
 * are formed with both functions and binned with angle_step (src/discretize.cpp:26-30). Returns the number of pairs
 * whose (k1, k2) bins differ; *n_pairs = pairs looked at, *n_bits = pairs whose angle bits differ at all, *max_abs =
 * largest |difference| of an angle; up to cap differing pairs are written to differing_ij. Test infrastructure. */
uint64_t tmo_atan2_census(const float* surfels, uint32_t n, const uint32_t* subset, uint32_t n_subset, const uint32_t* ij,
                          uint64_t n_ij, float lower_bound, float upper_bound, float angle_step, uint32_t* differing_ij,
                          uint32_t cap, uint64_t* n_pairs, uint64_t* n_bits, float* max_abs) {
    uint64_t looked = 0, bins = 0, bits = 0;
    float worst = 0.f;
    auto one = [&](uint32_t i, uint32_t j) {
        const float* s0 = surfels + (size_t)i * 12;
        const float* s1 = surfels + (size_t)j * 12;
        v3 p0 = s_pos(s0), p1 = s_pos(s1), t0 = s_tgt(s0), t1 = s_tgt(s1);
        v3 d0 = sub(p1, p0);
        const float y1 = norm(cross(d0, t0)), x1 = fabsf(dot(d0, t0)), y2 = norm(cross(d0, t1)), x2 = fabsf(dot(d0, t1));
        const float a1 = tmo_atan2f(y1, x1), a2 = tmo_atan2f(y2, x2);
        const float b1 = atan2f(y1, x1), b2 = atan2f(y2, x2); /* glibc */
        ++looked;
        if (memcmp(&a1, &b1, 4) || memcmp(&a2, &b2, 4)) ++bits;
        worst = std::max(worst, std::max(fabsf(a1 - b1), fabsf(a2 - b2)));
        if (tmo_discretize_step(a1, angle_step) != tmo_discretize_step(b1, angle_step) ||
            tmo_discretize_step(a2, angle_step) != tmo_discretize_step(b2, angle_step)) {
            if (differing_ij && bins < cap) {
                differing_ij[2 * bins] = i;
                differing_ij[2 * bins + 1] = j;
            }
            ++bins;
        }
    };
    if (ij) {
        for (uint64_t k = 0; k < n_ij; ++k)
            if (ij[2 * k] < n && ij[2 * k + 1] < n) one(ij[2 * k], ij[2 * k + 1]);
    } else {
        for (uint32_t a = 0; a < n_subset; ++a)
            for (uint32_t b = 0; b < n_subset; ++b) {
                const uint32_t i = subset[a], j = subset[b];
                if (i == j) continue;
                const float* si = surfels + (size_t)i * 12;
                const float* sj = surfels + (size_t)j * 12;
                v3 d1 = sub(s_pos(sj), s_pos(si));
                const float dist1 = norm(d1);
                d1 = divs(d1, dist1);
                if (dist1 < lower_bound || dist1 > upper_bound) continue;
                if (1.f - fabsf(dot(d1, s_tgt(si))) < 0.01f) continue;
                one(i, j);
            }
    }
    if (n_pairs) *n_pairs = looked;
    if (n_bits) *n_bits = bits;
    if (max_abs) *max_abs = worst;
    return bins;
}

/* the early-drop bound (include/impl/scene.hpp:493-498) as restated against its literal transcription over a grid
 * of (tried, |subset|, corrs): returns the number of points where the two differ (first few written to bad3) */
uint64_t tmo_early_drop_compare(uint32_t subset_max, uint32_t step_subset, uint32_t step_tried, uint32_t step_corrs,
                                uint64_t* n_points, uint32_t* bad3, uint32_t cap) {
    uint64_t looked = 0, bad = 0;
    for (uint32_t S = 1; S <= subset_max; S += step_subset)
        for (uint32_t tried = 1; tried <= S; tried += step_tried)
            for (uint32_t corrs = 0; corrs <= tried; corrs += step_corrs) {
                ++looked;
                if (tmo_early_drop_upper(tried, S, corrs) != tmo_early_drop_upper_literal(tried, S, corrs)) {
                    if (bad3 && bad < cap) {
                        bad3[3 * bad] = tried;
                        bad3[3 * bad + 1] = S;
                        bad3[3 * bad + 2] = corrs;
                    }
                    ++bad;
                }
            }
    if (n_points) *n_points = looked;
    return bad;
}

/* a3: ref: include/impl/feature.hpp:38-46. distance_step_count is a float passed to a uint32_t
 * parameter (implicit truncation); dim 3 is quantised with the dim-0 bounds (:44). */
void tmo_discretize_feature(const float f[4], const float bmin[4], const float bmax[4],
                            float distance_step_count, float angle_step, uint32_t key[4]) {
    uint32_t steps = (uint32_t)distance_step_count;
    float diag0 = bmax[0] - bmin[0];
    key[0] = tmo_discretize_range(f[0], bmin[0], diag0, steps);
    key[1] = tmo_discretize_step(f[1], angle_step);
    key[2] = tmo_discretize_step(f[2], angle_step);
    key[3] = tmo_discretize_range(f[3], bmin[0], diag0, steps);
}

/* a5: ref: include/impl/feature.hpp:50-88 */
int tmo_valid(const float f[4], const float bmin[4], const float bmax[4]) {
    if (f[0] < bmin[0] || f[0] > bmax[0]) return 0;
    float pi = (float)M_PI;
    bool angle_ok = (f[1] >= 0.f && f[1] <= pi) && (f[2] >= 0.f && f[2] <= pi);
    return angle_ok ? 1 : 0;
}

/* a5: ref: include/impl/feature.hpp:91-114 (only dims 0 and 3 are rescaled) */
void tmo_valid_bounds(const float bmin[4], const float bmax[4], float min_rel, float max_rel,
                      float out_min[4], float out_max[4]) {
    for (int d = 0; d < 4; ++d) {
        out_min[d] = bmin[d];
        out_max[d] = bmax[d];
    }
    float d0 = bmax[0] - bmin[0], d3 = bmax[3] - bmin[3];
    out_min[0] = bmin[0] + min_rel * d0;
    out_max[0] = bmin[0] + max_rel * d0;
    out_min[3] = bmin[3] + min_rel * d3;
    out_max[3] = bmin[3] + max_rel * d3;
}

/* a10: ref: include/impl/scene.hpp:538-567. base = [u | v | normalize(u x v)] (columns);
 * R = base_b * base_a^-1 with Eigen's fixed 3x3 inverse = cofactor matrix * (1/det);
 * t = o_b - R*o_a. Maps scene -> model. */
void tmo_base_transform(const float src_i[3], const float src_j[3], const float src_t[3],
                        const float tgt_i[3], const float tgt_j[3], const float tgt_t[3],
                        float T[16]) {
    v3 o_a = {src_i[0], src_i[1], src_i[2]};
    v3 o_b = {tgt_i[0], tgt_i[1], tgt_i[2]};
    v3 u_a = sub(v3{src_j[0], src_j[1], src_j[2]}, o_a);
    v3 u_b = sub(v3{tgt_j[0], tgt_j[1], tgt_j[2]}, o_b);
    v3 v_a = {src_t[0], src_t[1], src_t[2]};
    v3 v_b = {tgt_t[0], tgt_t[1], tgt_t[2]};
    u_a = normalized(u_a);
    u_b = normalized(u_b);
    v_a = sub(v_a, mul(u_a, dot(v_a, u_a)));
    v_b = sub(v_b, mul(u_b, dot(v_b, u_b)));
    v_a = normalized(v_a);
    v_b = normalized(v_b);
    v3 w_a = normalized(cross(u_a, v_a));
    v3 w_b = normalized(cross(u_b, v_b));
    /* A[r][c], columns u,v,w */
    float A[3][3] = {{u_a.x, v_a.x, w_a.x}, {u_a.y, v_a.y, w_a.y}, {u_a.z, v_a.z, w_a.z}};
    float B[3][3] = {{u_b.x, v_b.x, w_b.x}, {u_b.y, v_b.y, w_b.y}, {u_b.z, v_b.z, w_b.z}};
    /* cofactor(i,j) = A[i1][j1]*A[i2][j2] - A[i1][j2]*A[i2][j1], i1=(i+1)%3, i2=(i+2)%3 */
    auto cof = [&](int i, int j) {
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return A[i1][j1] * A[i2][j2] - A[i1][j2] * A[i2][j1];
    };
    float c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    float det = (c00 * A[0][0] + c10 * A[1][0]) + c20 * A[2][0];
    float invdet = 1.0f / det;
    float Ai[3][3];
    /* inverse(r,c) = cofactor(c,r) * invdet */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Ai[r][c] = cof(c, r) * invdet;
    float R[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            R[r][c] = (B[r][0] * Ai[0][c] + B[r][1] * Ai[1][c]) + B[r][2] * Ai[2][c];
    float Ro[3];
    for (int r = 0; r < 3; ++r) Ro[r] = (R[r][0] * o_a.x + R[r][1] * o_a.y) + R[r][2] * o_a.z;
    for (int i = 0; i < 16; ++i) T[i] = 0.f;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[c * 4 + r] = R[r][c];
    T[12] = o_b.x - Ro[0];
    T[13] = o_b.y - Ro[1];
    T[14] = o_b.z - Ro[2];
    T[15] = 1.f;
}

/* Q8 positive form of ref: include/impl/scene.hpp:493-498 */
uint32_t tmo_early_drop_upper(uint32_t tried, uint32_t subset_size, uint32_t corrs) {
    double N = 2.0 + tried, X = 2.0 + subset_size, C = 1.0 + corrs;
    double tmp = std::sqrt((X * C * (X - N) * (N - C)) / (N + 1.0));
    double q = (X * C + tmp) / N;
    uint32_t t = (uint32_t)q; /* q >= 0 */
    return t - 1u;            /* wraps to 0xFFFFFFFF for t == 0, as the literal form does */
}
/* literal transcription, kept to pin the positive form against the expression as written */
uint32_t tmo_early_drop_upper_literal(uint32_t tried, uint32_t subset_size, uint32_t corrs) {
    double N = -2.0 - tried;
    double x = -2.0 - (double)subset_size;
    double n = -1.0 - (double)corrs;
    double tmp = std::sqrt((x * n * (N - x) * (N - n)) / (N - 1.0));
    /* negative double -> uint32_t is UB; x86-64 g++ emits cvttsd2si r64 and keeps the low 32 bits.
     * Written here through int64_t so that the behaviour is defined and identical. */
    uint32_t inner = (uint32_t)(int64_t)((x * n + tmp) / N);
    uint32_t upper = (uint32_t)(int64_t)(-1.0 - (double)inner);
    return upper;
}
/* ref: include/impl/scene.hpp:422-426 */
void tmo_early_drop_checkpoints(uint32_t subset_size, uint32_t tests[18]) {
    constexpr float step_size = 0.05f;
    uint32_t n = (uint32_t)(1.f / step_size) - 2; /* 18 */
    for (uint32_t i = 0; i < n; ++i) tests[i] = (uint32_t)(step_size * (i + 1) * subset_size);
}

}  // extern "C"

/* =============================================================================================
 * model
 * =========================================================================================== */
namespace {

struct KeyHash {
    size_t operator()(const std::array<uint32_t, 4>& k) const { return tmo_murmur_key(k.data(), 4); }
};
typedef std::unordered_multimap<std::array<uint32_t, 4>, std::pair<uint32_t, uint32_t>, KeyHash>
    hash_map_t; /* ref: include/model:25-26 */

struct KdNode {
    int left, right;   /* children or -1 */
    int begin, end;    /* leaf range into order[] */
    int dim;
    float split;
};

struct KdTree {
    std::vector<KdNode> nodes;
    std::vector<uint32_t> order;
    const float* pts = nullptr; /* surfels */

    float coord(uint32_t i, int d) const { return pts[(size_t)i * 12 + d]; }

    int build(int b, int e) {
        KdNode nd;
        nd.left = nd.right = -1;
        nd.begin = b;
        nd.end = e;
        nd.dim = 0;
        nd.split = 0.f;
        int id = (int)nodes.size();
        nodes.push_back(nd);
        if (e - b <= 8) return id;
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = b; i < e; ++i)
            for (int d = 0; d < 3; ++d) {
                float c = coord(order[i], d);
                mn[d] = std::min(mn[d], c);
                mx[d] = std::max(mx[d], c);
            }
        int dim = 0;
        if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
        if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        if (!(mx[dim] - mn[dim] > 0.f)) return id; /* all coincident: keep as leaf */
        int mid = (b + e) / 2;
        std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e,
                         [&](uint32_t a, uint32_t c) { return coord(a, dim) < coord(c, dim); });
        float split = coord(order[mid], dim);
        int l = build(b, mid);
        int r = build(mid, e);
        nodes[id].left = l;
        nodes[id].right = r;
        nodes[id].dim = dim;
        nodes[id].split = split;
        return id;
    }

    void init(const float* surfels, const std::vector<uint32_t>& idx) {
        pts = surfels;
        order = idx;
        nodes.clear();
        if (!order.empty()) build(0, (int)order.size());
    }

    /* exact NN under d2 = (dx*dx + dy*dy) + dz*dz (fp32), ties -> lowest point index */
    void search(int id, const float q[3], float& best, uint32_t& best_i) const {
        const KdNode& nd = nodes[id];
        if (nd.left < 0) {
            for (int i = nd.begin; i < nd.end; ++i) {
                uint32_t p = order[i];
                float dx = coord(p, 0) - q[0], dy = coord(p, 1) - q[1], dz = coord(p, 2) - q[2];
                float d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 < best || (d2 == best && p < best_i)) {
                    best = d2;
                    best_i = p;
                }
            }
            return;
        }
        float diff = q[nd.dim] - nd.split;
        int near = diff < 0.f ? nd.left : nd.right;
        int far = diff < 0.f ? nd.right : nd.left;
        search(near, q, best, best_i);
        /* rounding is monotone, so fl(diff^2) <= fl(d2) for every point of the far side:
         * pruning on a strict '>' cannot lose an equal-distance lower index */
        if (!(diff * diff > best)) search(far, q, best, best_i);
    }
};

}  // namespace

struct tmo_model {
    std::vector<float> cloud; /* n*12 */
    uint32_t n = 0;
    float resolution = 0.f;
    float distance_step_count = 0.f, angle_step = 0.f;
    bool voxels_ready = false, init = false;
    float diameter = 0.f;
    int extents[3] = {0, 0, 0};
    int margin = 5;
    float to_voxel[16];
    std::vector<uint32_t> voxels;
    std::vector<uint32_t> grid_subset; /* subset handed to init(): bbox / diameter / grid over it; empty = all */
    std::vector<uint32_t> subset;
    float bmin[4], bmax[4];
    uint64_t valid_count = 0;
    hash_map_t map;
    std::vector<uint8_t> is_tangent; /* |tangent| > 0.7 per model point, ref: scene.hpp:469-470 */
    int metric_kind = 0;      /* traits whose intrinsic_distance measures query <-> model point */
    float metric_radius = 0.f; /* cylinder radius for the wrap-around */
};

namespace {

bool finite3(v3 a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }

/* ref: include/impl/model.hpp:24-61. bbox over the finite points of the (default: whole) cloud. */
int model_setup_grid(tmo_model* m) {
    const float* c = m->cloud.data();
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
                   std::numeric_limits<float>::max()};
    float mx[3] = {std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(),
                   std::numeric_limits<float>::lowest()};
    uint32_t kept = 0;
    /* :17-22: an empty subset stands for the whole cloud; :33-37: bbox over the (finite) subset */
    const uint32_t n_sel = m->grid_subset.empty() ? m->n : (uint32_t)m->grid_subset.size();
    for (uint32_t k = 0; k < n_sel; ++k) {
        const uint32_t i = m->grid_subset.empty() ? k : m->grid_subset[k];
        const float* s = c + (size_t)i * 12;
        if (!(finite3(s_pos(s)) && finite3(s_nrm(s)) && finite3(s_tgt(s)))) continue; /* :24-30 */
        ++kept;
        for (int d = 0; d < 3; ++d) {
            mn[d] = std::min(mn[d], s[d]);
            mx[d] = std::max(mx[d], s[d]);
        }
    }
    if (!kept) return -2;
    v3 range = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    m->diameter = norm(range); /* :39 */
    float cell = 0.5f * m->resolution;
    float ext[3] = {std::max(range.x / cell, 1.f), std::max(range.y / cell, 1.f),
                    std::max(range.z / cell, 1.f)}; /* :45-46 */
    m->margin = 5;
    float two_margin = 2.f * (float)m->margin;
    for (int d = 0; d < 3; ++d) m->extents[d] = (int)(ext[d] + two_margin); /* :50 */
    float rg[3] = {range.x, range.y, range.z};
    float scale[3];
    for (int d = 0; d < 3; ++d) scale[d] = rg[d] < 1e-5f ? 1.f : ext[d] / rg[d]; /* :52-55 */
    for (int i = 0; i < 16; ++i) m->to_voxel[i] = 0.f;
    m->to_voxel[0] = scale[0];
    m->to_voxel[5] = scale[1];
    m->to_voxel[10] = scale[2];
    m->to_voxel[15] = 1.f;
    for (int d = 0; d < 3; ++d) {
        float t = scale[d] * (-mn[d]) + (float)m->margin; /* :57-58 */
        t -= 0.5f;                                         /* :60 */
        m->to_voxel[12 + d] = t;
    }
    return 0;
}

/* voxel centre: inverse of the diagonal-affine to_voxel applied to (i,j,k,1) (ref: model.hpp:63,87).
 * Eigen's general 4x4 inverse is unpinnable; defined here as inv_s = 1/s, inv_t = -(t*inv_s),
 * centre = inv_s*i + inv_t. */
inline void voxel_centre(const tmo_model* m, int i, int j, int k, float q[3]) {
    int ijk[3] = {i, j, k};
    for (int d = 0; d < 3; ++d) {
        float inv_s = 1.0f / m->to_voxel[d * 5];
        float inv_t = -(m->to_voxel[12 + d] * inv_s);
        q[d] = inv_s * (float)ijk[d] + inv_t;
    }
}

template <typename F>
void parallel_for(int n, int threads, F&& fn) {
    if (threads <= 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<int> next(0);
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; ++t)
        ts.emplace_back([&]() {
            for (;;) {
                int i = next.fetch_add(1);
                if (i >= n) break;
                fn(i);
            }
        });
    for (auto& t : ts) t.join();
}

}  // namespace

extern "C" {

/* ref: include/impl/pointcloud.hpp:66-82 + include/common:104-115. nearestKSearch(pnt, 2) returns the
 * point itself and its nearest other point (FLANN: exact L2); exhaustive scan here. The running mean is
 * taken in f32 in point order, as `average` does. */
float tmo_cloud_resolution(const float* surfels, uint32_t n, float* nn_dist, int threads) {
    std::vector<float> dist(n, 0.f);
    parallel_for((int)n, threads, [&](int i) {
        const float* q = surfels + (size_t)i * 12;
        if (!finite3(s_pos(q))) return;
        float b1 = std::numeric_limits<float>::infinity(), b2 = b1;
        for (uint32_t j = 0; j < n; ++j) {
            const float* p = surfels + (size_t)j * 12;
            if (!finite3(s_pos(p))) continue;
            float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
            float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < b1) {
                b2 = b1;
                b1 = d2;
            } else if (d2 < b2) {
                b2 = d2;
            }
        }
        dist[i] = sqrtf(b2);
    });
    float accum = 0.f;
    uint32_t i = 0;
    for (uint32_t k = 0; k < n; ++k) {
        if (!finite3(s_pos(surfels + (size_t)k * 12))) continue;
        accum = accum + (dist[k] - accum) / (float)(++i);
    }
    if (nn_dist) memcpy(nn_dist, dist.data(), (size_t)n * 4);
    return accum;
}

/* ref: include/impl/pointcloud.hpp:3-44 (principal_curvatures) + :200-204 (curvature(k, idx)).
 * knn_inclusive: the k nearest points, the query itself included, by distance ascending (ties by index:
 * FLANN's order on ties is unpinned). pcl::eigen33 / computeCorrespondingEigenVector are third-party:
 * defined as a 10-sweep cyclic Jacobi iteration in f32. Exhaustive neighbour search. */
static void oracle_jacobi3(const float a_in[6], float ev[3], float vmax[3]) {
    float a[3][3] = {{a_in[0], a_in[1], a_in[2]}, {a_in[1], a_in[3], a_in[4]}, {a_in[2], a_in[4], a_in[5]}};
    float v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int sweep = 0; sweep < 10; ++sweep)
        for (int pq = 0; pq < 3; ++pq) {
            int p = P[pq], q = Q[pq], r = 3 - p - q;
            float apq = a[p][q];
            if (apq == 0.f) continue;
            float theta = (a[q][q] - a[p][p]) / (2.f * apq);
            float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
            float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
            a[p][p] = a[p][p] - t * apq;
            a[q][q] = a[q][q] + t * apq;
            a[p][q] = a[q][p] = 0.f;
            float arp = a[r][p], arq = a[r][q];
            a[r][p] = a[p][r] = c * arp - s * arq;
            a[r][q] = a[q][r] = s * arp + c * arq;
            for (int k = 0; k < 3; ++k) {
                float vkp = v[k][p], vkq = v[k][q];
                v[k][p] = c * vkp - s * vkq;
                v[k][q] = s * vkp + c * vkq;
            }
        }
    int o[3] = {0, 1, 2};
    if (a[o[0]][o[0]] > a[o[1]][o[1]]) std::swap(o[0], o[1]);
    if (a[o[1]][o[1]] > a[o[2]][o[2]]) std::swap(o[1], o[2]);
    if (a[o[0]][o[0]] > a[o[1]][o[1]]) std::swap(o[0], o[1]);
    for (int k = 0; k < 3; ++k) {
        ev[k] = a[o[k]][o[k]];
        vmax[k] = v[k][o[2]];
    }
}

void tmo_cloud_curvature(const float* surfels, uint32_t n, uint32_t k, float* curv5, int threads) {
    parallel_for((int)n, threads, [&](int i) {
        float* out = curv5 + (size_t)i * 5;
        for (int c = 0; c < 5; ++c) out[c] = 0.f;
        const float* q = surfels + (size_t)i * 12;
        if (!finite3(s_pos(q))) return;
        std::vector<std::pair<float, uint32_t>> nb;
        nb.reserve(n);
        for (uint32_t j = 0; j < n; ++j) {
            const float* p = surfels + (size_t)j * 12;
            if (!finite3(s_pos(p))) continue;
            float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
            nb.push_back({(dx * dx + dy * dy) + dz * dz, j});
        }
        uint32_t kk = std::min<uint32_t>(k, (uint32_t)nb.size());
        std::partial_sort(nb.begin(), nb.begin() + kk, nb.end()); /* (d2, index) ascending */
        v3 n0 = s_nrm(q);
        float n0a[3] = {n0.x, n0.y, n0.z};
        float M[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) M[r][c] = (r == c ? 1.f : 0.f) - n0a[r] * n0a[c]; /* :8-10 */
        float cen[3] = {0, 0, 0};
        std::vector<std::array<float, 3>> proj(kk);
        for (uint32_t j = 0; j < kk; ++j) { /* :13-16 */
            v3 nj = s_nrm(surfels + (size_t)nb[j].second * 12);
            for (int r = 0; r < 3; ++r) {
                proj[j][r] = (M[r][0] * nj.x + M[r][1] * nj.y) + M[r][2] * nj.z;
                cen[r] = cen[r] + (proj[j][r] - cen[r]) / (float)(j + 1);
            }
        }
        float cov[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t j = 0; j < kk; ++j) { /* :19-35 */
            float d[3] = {proj[j][0] - cen[0], proj[j][1] - cen[1], proj[j][2] - cen[2]};
            cov[0] += d[0] * d[0];
            cov[1] += d[0] * d[1];
            cov[2] += d[0] * d[2];
            cov[3] += d[1] * d[1];
            cov[4] += d[1] * d[2];
            cov[5] += d[2] * d[2];
        }
        float ev[3], vmax[3];
        oracle_jacobi3(cov, ev, vmax);
        float area_inv = 1.0f / (float)kk; /* :42 */
        out[0] = vmax[0];
        out[1] = vmax[1];
        out[2] = vmax[2];
        out[3] = ev[1] * area_inv;
        out[4] = ev[2] * area_inv;
    });
}

/* pointcloud<Point>::knn_inclusive(k, point) (include/impl/pointcloud.hpp:138-146) and
 * radius_search_inclusive(r, point) (:168-176) restated exhaustively: every finite point of the cloud (or of
 * `subset`, set_indices / build_tree_ :224-240) against the query, order (squared distance, index) ascending —
 * FLANN leaves ties open, this is the order the restatement defines. Distances are squared, as PCL returns. */
static void cloud_candidates(const float* surfels, uint32_t n, const uint32_t* subset, uint32_t n_subset,
                             const float q[3], std::vector<std::pair<float, uint32_t>>& nb) {
    nb.clear();
    if (!finite3(v3{q[0], q[1], q[2]})) return;
    auto add = [&](uint32_t j) {
        const float* p = surfels + (size_t)j * 12;
        if (!finite3(s_pos(p))) return;
        float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
        nb.push_back({(dx * dx + dy * dy) + dz * dz, j});
    };
    if (n_subset)
        for (uint32_t k = 0; k < n_subset; ++k) add(subset[k]);
    else
        for (uint32_t j = 0; j < n; ++j) add(j);
}

uint32_t tmo_cloud_knn(const float* surfels, uint32_t n, const uint32_t* subset, uint32_t n_subset,
                       const float query[3], uint32_t k, int32_t* indices, float* sq_dists) {
    std::vector<std::pair<float, uint32_t>> nb;
    cloud_candidates(surfels, n, subset, n_subset, query, nb);
    const uint32_t kk = std::min<uint32_t>(k, (uint32_t)nb.size());
    std::partial_sort(nb.begin(), nb.begin() + kk, nb.end());
    for (uint32_t j = 0; j < kk; ++j) {
        indices[j] = (int32_t)nb[j].second;
        sq_dists[j] = nb[j].first;
    }
    return kk;
}

uint32_t tmo_cloud_radius_search(const float* surfels, uint32_t n, const uint32_t* subset, uint32_t n_subset,
                                 const float query[3], float radius, int32_t* indices, float* sq_dists,
                                 uint32_t cap) {
    std::vector<std::pair<float, uint32_t>> nb, in;
    cloud_candidates(surfels, n, subset, n_subset, query, nb);
    const float r2 = radius * radius;
    for (const auto& c : nb)
        if (c.first <= r2) in.push_back(c);
    std::sort(in.begin(), in.end());
    for (uint32_t j = 0; j < in.size() && j < cap; ++j) {
        if (indices) indices[j] = (int32_t)in[j].second;
        if (sq_dists) sq_dists[j] = in[j].first;
    }
    return (uint32_t)in.size();
}

/* principal_curvatures(cloud, p_idx, indices) (include/impl/pointcloud.hpp:3-44) over an explicit neighbour list:
 * what curvature(float r, idx) evaluates on radius_search_inclusive(r, idx).first (:206-209) */
void tmo_cloud_curvature_of(const float* surfels, uint32_t p_idx, const int32_t* indices, uint32_t n_idx,
                            float curv5[5]) {
    for (int c = 0; c < 5; ++c) curv5[c] = 0.f;
    if (n_idx == 0) return;
    v3 n0 = s_nrm(surfels + (size_t)p_idx * 12);
    float n0a[3] = {n0.x, n0.y, n0.z};
    float M[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[r][c] = (r == c ? 1.f : 0.f) - n0a[r] * n0a[c]; /* :8-10 */
    float cen[3] = {0, 0, 0};
    std::vector<std::array<float, 3>> proj(n_idx);
    for (uint32_t j = 0; j < n_idx; ++j) { /* :13-16 */
        v3 nj = s_nrm(surfels + (size_t)indices[j] * 12);
        for (int r = 0; r < 3; ++r) {
            proj[j][r] = (M[r][0] * nj.x + M[r][1] * nj.y) + M[r][2] * nj.z;
            cen[r] = cen[r] + (proj[j][r] - cen[r]) / (float)(j + 1);
        }
    }
    float cov[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j = 0; j < n_idx; ++j) { /* :19-35 */
        float d[3] = {proj[j][0] - cen[0], proj[j][1] - cen[1], proj[j][2] - cen[2]};
        cov[0] += d[0] * d[0];
        cov[1] += d[0] * d[1];
        cov[2] += d[0] * d[2];
        cov[3] += d[1] * d[1];
        cov[4] += d[1] * d[2];
        cov[5] += d[2] * d[2];
    }
    float ev[3], vmax[3];
    oracle_jacobi3(cov, ev, vmax);
    float area_inv = 1.0f / (float)n_idx; /* :42 */
    curv5[0] = vmax[0];
    curv5[1] = vmax[1];
    curv5[2] = vmax[2];
    curv5[3] = ev[1] * area_inv;
    curv5[4] = ev[2] * area_inv;
}

tmo_model* tmo_model_create(const float* surfels, uint32_t n, float resolution,
                            float distance_step_count, float angle_step) {
    tmo_model* m = new tmo_model();
    m->cloud.assign(surfels, surfels + (size_t)n * 12);
    m->n = n;
    m->resolution = resolution;
    m->distance_step_count = distance_step_count;
    m->angle_step = angle_step;
    m->is_tangent.resize(n);
    for (uint32_t i = 0; i < n; ++i)
        m->is_tangent[i] = norm(s_tgt(surfels + (size_t)i * 12)) > 0.7f ? 1 : 0;
    for (int d = 0; d < 4; ++d) {
        m->bmin[d] = std::numeric_limits<float>::max();
        m->bmax[d] = std::numeric_limits<float>::lowest();
    }
    return m;
}
void tmo_model_destroy(tmo_model* m) { delete m; }

/* a6: ref: include/impl/model.hpp:81-94 — every voxel stores the nearest cloud point (whole cloud,
 * :88) to its centre. PCL/FLANN tie-breaking is unpinned; defined here as lowest index. */
static int build_voxels_impl(tmo_model* m, int threads, bool brute) {
    int rc = model_setup_grid(m);
    if (rc) return rc;
    const int ex = m->extents[0], ey = m->extents[1], ez = m->extents[2];
    uint64_t count = (uint64_t)ex * ey * ez;
    if (count > 0xFFFFFFFFull) return -3; /* reference's uint32_t voxel_count overflows (:65) */
    m->voxels.assign(count, 0u);
    std::vector<uint32_t> idx;
    idx.reserve(m->n);
    for (uint32_t i = 0; i < m->n; ++i)
        if (finite3(s_pos(m->cloud.data() + (size_t)i * 12))) idx.push_back(i);
    KdTree tree;
    if (!brute) tree.init(m->cloud.data(), idx);
    const float* c = m->cloud.data();
    parallel_for(ez, threads, [&](int k) {
        for (int j = 0; j < ey; ++j)
            for (int i = 0; i < ex; ++i) {
                float q[3];
                voxel_centre(m, i, j, k, q);
                float best = std::numeric_limits<float>::infinity();
                uint32_t best_i = 0xFFFFFFFFu;
                if (brute) {
                    for (uint32_t p : idx) {
                        float dx = c[(size_t)p * 12] - q[0], dy = c[(size_t)p * 12 + 1] - q[1],
                              dz = c[(size_t)p * 12 + 2] - q[2];
                        float d2 = (dx * dx + dy * dy) + dz * dz;
                        if (d2 < best) {
                            best = d2;
                            best_i = p;
                        }
                    }
                } else {
                    tree.search(0, q, best, best_i);
                }
                m->voxels[(size_t)k * ex * ey + (size_t)j * ex + i] = best_i; /* :84 */
            }
    });
    m->voxels_ready = true;
    return 0;
}
int tmo_model_set_grid_subset(tmo_model* m, const uint32_t* subset, uint32_t n) {
    m->grid_subset.assign(subset, subset + n);
    m->voxels_ready = false;
    return model_setup_grid(m);
}
int tmo_model_build_voxels(tmo_model* m, int threads) { return build_voxels_impl(m, threads, false); }
int tmo_model_build_voxels_bruteforce(tmo_model* m, int threads) {
    return build_voxels_impl(m, threads, true);
}
int tmo_model_set_voxels(tmo_model* m, const uint32_t* voxels, uint64_t count) {
    int rc = model_setup_grid(m);
    if (rc) return rc;
    uint64_t expect = (uint64_t)m->extents[0] * m->extents[1] * m->extents[2];
    if (expect != count) return -4;
    m->voxels.assign(voxels, voxels + count);
    m->voxels_ready = true;
    return 0;
}

/* a7: ref: include/impl/model.hpp:96-149. `subset` is the already tangent/curvature-filtered
 * subset_ (:96-99; curvature is a "next" row, supplied by the caller). Pairs are enumerated
 * i-major/j-minor over subset positions (:100); pass 1 extends the feature bounds (:103-118),
 * valid_bounds(.,.,.,0,1) (:122), pass 2 inserts {key,(i,j)} (:125-149). The /tmp/out0.dat
 * histogram dump (:151-164) is not replicated. */
int tmo_model_build_pairs(tmo_model* m, const uint32_t* subset, uint32_t n_subset,
                          float min_diameter_factor, float max_diameter_factor) {
    if (m->diameter == 0.f) {
        int rc = model_setup_grid(m);
        if (rc) return rc;
    }
    m->subset.assign(subset, subset + n_subset);
    m->map.clear();
    const float* c = m->cloud.data();
    float lower_bound = m->diameter * min_diameter_factor;
    float upper_bound = m->diameter * max_diameter_factor;
    float bmin[4], bmax[4];
    for (int d = 0; d < 4; ++d) {
        bmin[d] = std::numeric_limits<float>::max();
        bmax[d] = std::numeric_limits<float>::lowest();
    }
    uint64_t valid_count = 0;
    auto passes_filters = [&](uint32_t i, uint32_t j) {
        if (i == j) return false;
        const float* si = c + (size_t)i * 12;
        const float* sj = c + (size_t)j * 12;
        v3 d1 = sub(s_pos(sj), s_pos(si));
        float dist1 = norm(d1);
        d1 = divs(d1, dist1);
        if (dist1 < lower_bound || dist1 > upper_bound) return false;
        if (1.f - fabsf(dot(d1, s_tgt(si))) < 0.01f) return false;
        return true;
    };
    for (uint32_t a = 0; a < n_subset; ++a)
        for (uint32_t b = 0; b < n_subset; ++b) {
            uint32_t i = subset[a], j = subset[b];
            if (!passes_filters(i, j)) continue;
            float f[4];
            tmo_feature(c + (size_t)i * 12, c + (size_t)j * 12, f);
            for (int d = 0; d < 4; ++d) {
                bmin[d] = std::min(bmin[d], f[d]);
                bmax[d] = std::max(bmax[d], f[d]);
            }
            ++valid_count;
        }
    m->valid_count = valid_count;
    if (!valid_count) return -5; /* reference dereferences max_element of an empty map here */
    tmo_valid_bounds(bmin, bmax, 0.0f, 1.f, m->bmin, m->bmax);
    for (uint32_t a = 0; a < n_subset; ++a)
        for (uint32_t b = 0; b < n_subset; ++b) {
            uint32_t i = subset[a], j = subset[b];
            if (!passes_filters(i, j)) continue;
            float f[4];
            tmo_feature(c + (size_t)i * 12, c + (size_t)j * 12, f);
            if (!tmo_valid(f, m->bmin, m->bmax)) continue;
            std::array<uint32_t, 4> key;
            tmo_discretize_feature(f, m->bmin, m->bmax, m->distance_step_count, m->angle_step,
                                   key.data());
            m->map.insert({key, {i, j}});
        }
    m->init = true;
    return 0;
}

float tmo_model_diameter(const tmo_model* m) { return m->diameter; }
void tmo_model_set_metric(tmo_model* m, int kind, float radius) {
    m->metric_kind = kind;
    m->metric_radius = radius;
}
uint32_t tmo_model_size(const tmo_model* m) { return m->n; }
uint32_t tmo_model_point_count(const tmo_model* m) { return (uint32_t)m->subset.size(); }
void tmo_model_get_grid(const tmo_model* m, int extents[3], float to_voxel16[16]) {
    for (int d = 0; d < 3; ++d) extents[d] = m->extents[d];
    for (int i = 0; i < 16; ++i) to_voxel16[i] = m->to_voxel[i];
}
uint64_t tmo_model_voxel_count(const tmo_model* m) { return m->voxels.size(); }
const uint32_t* tmo_model_voxels(const tmo_model* m) { return m->voxels.data(); }
void tmo_model_get_feature_bounds(const tmo_model* m, float bmin[4], float bmax[4]) {
    for (int d = 0; d < 4; ++d) {
        bmin[d] = m->bmin[d];
        bmax[d] = m->bmax[d];
    }
}
uint64_t tmo_model_pair_count(const tmo_model* m) { return m->map.size(); }
uint64_t tmo_model_valid_count(const tmo_model* m) { return m->valid_count; }

/* a8: ref: include/impl/model.hpp:169-178; entries come back in libstdc++ unordered_multimap
 * equal_range order (this IS that container, so the order is the reference's on this platform). */
int tmo_model_query_key(const tmo_model* m, const uint32_t key[4], uint32_t* pairs_ij, uint32_t cap,
                        uint32_t* count_total) {
    if (!m->init) return -1;
    std::array<uint32_t, 4> k = {key[0], key[1], key[2], key[3]};
    auto range = m->map.equal_range(k);
    uint32_t n = 0;
    for (auto it = range.first; it != range.second; ++it, ++n) {
        if (n < cap && pairs_ij) {
            pairs_ij[2 * n] = it->second.first;
            pairs_ij[2 * n + 1] = it->second.second;
        }
    }
    if (count_total) *count_total = n;
    return 0;
}
int tmo_model_query(const tmo_model* m, const float f[4], uint32_t* pairs_ij, uint32_t cap,
                    uint32_t* count_total) {
    if (!m->init) return -1;
    uint32_t key[4];
    tmo_discretize_feature(f, m->bmin, m->bmax, m->distance_step_count, m->angle_step, key);
    return tmo_model_query_key(m, key, pairs_ij, cap, count_total);
}
uint32_t tmo_model_table_keys(const tmo_model* m, uint32_t* keys4, uint32_t* counts, uint32_t cap) {
    std::map<std::array<uint32_t, 4>, uint32_t> hist;
    for (auto& kv : m->map) hist[kv.first] += 1;
    uint32_t n = 0;
    for (auto& kv : hist) {
        if (n < cap) {
            if (keys4) memcpy(keys4 + 4 * n, kv.first.data(), 16);
            if (counts) counts[n] = kv.second;
        }
        ++n;
    }
    return n;
}

/* a9: ref: include/impl/model.hpp:180-192. The int cast truncates toward zero (values in (-1,0)
 * land in cell 0); casts of non-finite / out-of-int-range values (UB in the reference; INT_MIN on
 * x86-64, i.e. rejected) are defined here as "outside". */
int tmo_model_voxel_query(const tmo_model* m, const float pos4[4], uint32_t* index) {
    float v[4];
    mat4_mul(m->to_voxel, pos4[0], pos4[1], pos4[2], pos4[3], v);
    int ijk[3];
    for (int d = 0; d < 3; ++d) {
        if (!(v[d] > -2147483648.f && v[d] < 2147483648.f)) return 0;
        ijk[d] = (int)v[d];
    }
    if (ijk[0] < 0 || ijk[1] < 0 || ijk[2] < 0 || ijk[0] >= m->extents[0] ||
        ijk[1] >= m->extents[1] || ijk[2] >= m->extents[2])
        return 0;
    size_t lin = (size_t)ijk[2] * m->extents[0] * m->extents[1] + (size_t)ijk[1] * m->extents[0] +
                 ijk[0];
    *index = m->voxels[lin];
    return 1;
}

}  // extern "C"

/* =============================================================================================
 * scene
 * =========================================================================================== */
struct tmo_scene {
    std::vector<float> cloud;
    uint32_t n = 0;
    std::vector<int32_t> mask;         /* ref: scene.hpp:41 */
    std::vector<int32_t> tangent_mask; /* ref: scene.hpp:42,50-53 (curvature test supplied by caller) */
};

extern "C" {

tmo_scene* tmo_scene_create(const float* surfels, uint32_t n, const int32_t* tangent_mask) {
    tmo_scene* s = new tmo_scene();
    s->cloud.assign(surfels, surfels + (size_t)n * 12);
    s->n = n;
    s->mask.assign(n, 0);
    s->tangent_mask.assign(tangent_mask, tangent_mask + n);
    return s;
}
void tmo_scene_destroy(tmo_scene* s) { delete s; }
void tmo_scene_set_mask(tmo_scene* s, const int32_t* mask) {
    if (mask)
        s->mask.assign(mask, mask + s->n);
    else
        s->mask.assign(s->n, 0);
}

/* a11/a14: ref: include/impl/scene.hpp:411-510 with the live constants of :11-28
 * (samples_on_all=true => considered_correspondence_ always true; naive_corrs=false;
 * weighted_distance=false; negative_scores=false; voxel_multiplicity=1). */
uint32_t tmo_project(const tmo_scene* s, const tmo_model* m, const int32_t* subset,
                     uint32_t n_subset, const float T[16], float accept_prob, float dist_thres,
                     int early_out, uint32_t* scene_corrs, uint32_t* model_corrs, double* score_out,
                     uint32_t* saved_out) {
    const uint32_t S = subset ? n_subset : s->n;
    float thres = dist_thres * m->resolution; /* :413 */
    uint32_t tried = 0;
    uint32_t next_test = 0;
    uint32_t tests[18];
    tmo_early_drop_checkpoints(S, tests);
    uint32_t saved = 0;
    double signed_score = 0;
    uint32_t n_corr = 0;
    const float* mc = m->cloud.data();
    float model_size_thresh = accept_prob * (float)m->n; /* :500 accept_prob * size() in float */
    for (uint32_t it = 0; it < S; ++it) {
        int idx = subset ? subset[it] : (int)it;
        ++tried;
        bool at_tail = false; /* `continue` at :434 and :458-460 bypasses the early-drop test */
        if (s->mask[idx] == 0) { /* :434 */
            const float* pnt = s->cloud.data() + (size_t)idx * 12;
            bool use_tangent = s->tangent_mask[idx] != 0;     /* :441 */
            v3 ref = use_tangent ? s_tgt(pnt) : s_nrm(pnt);   /* :442 */
            float pos[4];
            mat4_mul(T, pnt[0], pnt[1], pnt[2], 1.f, pos);    /* :444 */
            uint32_t mi;
            if (tmo_model_voxel_query(m, pos, &mi)) {         /* :455-460 */
                at_tail = true;
                ref = mat3_mul(T, ref);                       /* :461 */
                const float* mp = mc + (size_t)mi * 12;
                /* :464; for a model living in a projector's uvw space the distance is that traits'
                 * intrinsic_distance (remnant: include/impl/model.hpp:194-201 voxel_distance) */
                float dist;
                if (m->metric_kind == TMO_TRAITS_IDENTITY) {
                    dist = norm(sub(v3{pos[0], pos[1], pos[2]}, s_pos(mp)));
                } else {
                    float st[TMO_STATE_FLOATS] = {0};
                    st[0] = m->metric_radius;
                    dist = tmo_traits_intrinsic_distance(m->metric_kind, st, pos, mp);
                }
                if (!(dist > thres)) {                        /* :465 */
                    bool is_tangent = m->is_tangent[mi] != 0; /* :469-470 */
                    if (use_tangent == is_tangent) {          /* :472 */
                        v3 ref_n = is_tangent ? s_tgt(mp) : s_nrm(mp); /* :479-481 */
                        signed_score += (double)fabsf(dot(ref, ref_n)); /* :483 */
                        if (scene_corrs) scene_corrs[n_corr] = (uint32_t)idx;
                        if (model_corrs) model_corrs[n_corr] = mi;
                        ++n_corr;
                    }
                }
            }
        }
        /* :492-506 — the test sits at the loop tail: it runs only for unmasked points that got a
         * voxel hit, at the first such point with tried >= tests[next_test]. */
        if (at_tail && early_out && next_test < 18 && tried >= tests[next_test]) {
            uint32_t upper = tmo_early_drop_upper(tried, S, n_corr);
            if ((float)upper < model_size_thresh) { /* :500, u32 < float*size_t in float */
                saved = S - tried;
                if (score_out) *score_out = signed_score; /* raw partial sum, :502 */
                if (saved_out) *saved_out = saved;
                return n_corr;
            }
            ++next_test;
        }
    }
    if (score_out) *score_out = signed_score / (double)m->n; /* :509, :406-409 */
    if (saved_out) *saved_out = saved;
    return n_corr;
}

}  // extern "C"

extern "C" {

/* a12 inner body: ref: include/impl/scene.hpp:288-319 for an explicit, ordered list of scene (i,j)
 * pairs. Filters (:290-297): tangent_mask_[j], !mask_[j], i != j, squared-distance window against
 * squared bounds (:117-120), collinearity; feature/valid (:299-302); query (:304) truncated to
 * query_limit entries in container order (:307-310); base_transform_ (:315) with the tangent of
 * m_i only; optional force_up filter (:317-319). */
int tmo_generate_hypotheses(const tmo_scene* s, const tmo_model* m, const uint32_t* ij_pairs,
                            uint32_t n_pairs, float min_diameter_factor, float max_diameter_factor,
                            int force_up, int query_limit, float* poses16, uint32_t* provenance4,
                            uint32_t cap, uint32_t* count) {
    if (!m->init) return -1;
    float lower = m->diameter * min_diameter_factor;
    float upper = m->diameter * max_diameter_factor;
    lower *= lower;
    upper *= upper;
    uint32_t n_out = 0;
    std::vector<uint32_t> entries;
    const float* mc = m->cloud.data();
    for (uint32_t p = 0; p < n_pairs; ++p) {
        uint32_t i = ij_pairs[2 * p], j = ij_pairs[2 * p + 1];
        if (s->mask[i]) continue; /* :268 (fp_mask_ is orchestration state of the caller) */
        if (!s->tangent_mask[j] || s->mask[j] || i == j) continue; /* :290 */
        const float* p1 = s->cloud.data() + (size_t)i * 12;
        const float* p2 = s->cloud.data() + (size_t)j * 12;
        v3 d0 = sub(s_pos(p2), s_pos(p1));
        float sqn0 = sqn(d0);
        d0 = normalized(d0);
        if (sqn0 < lower || sqn0 > upper) continue;                     /* :296 */
        if (1.f - fabsf(dot(d0, s_tgt(p1))) < 0.01f) continue;          /* :297 */
        float f[4];
        tmo_feature(p1, p2, f);
        if (!tmo_valid(f, m->bmin, m->bmax)) continue;                  /* :300 */
        uint32_t total = 0;
        uint32_t key[4];
        tmo_discretize_feature(f, m->bmin, m->bmax, m->distance_step_count, m->angle_step, key);
        tmo_model_query_key(m, key, nullptr, 0, &total);
        uint32_t take = total;
        if (query_limit > 0 && take > (uint32_t)query_limit) take = (uint32_t)query_limit;
        entries.resize(2 * (size_t)take);
        tmo_model_query_key(m, key, entries.data(), take, &total);
        for (uint32_t e = 0; e < take; ++e) {
            uint32_t m_i = entries[2 * e], m_j = entries[2 * e + 1];
            const float* pmi = mc + (size_t)m_i * 12;
            const float* pmj = mc + (size_t)m_j * 12;
            float T[16];
            tmo_base_transform(p1, p2, p1 + 9, pmi, pmj, pmi + 9, T);   /* :315 */
            if (force_up && fabsf(1.f - T[10]) > 0.01f) continue;       /* :317, t(2,2) */
            if (n_out < cap) {
                if (poses16) memcpy(poses16 + (size_t)n_out * 16, T, 64);
                if (provenance4) {
                    provenance4[4 * (size_t)n_out + 0] = p;
                    provenance4[4 * (size_t)n_out + 1] = m_i;
                    provenance4[4 * (size_t)n_out + 2] = m_j;
                    provenance4[4 * (size_t)n_out + 3] = e;
                }
            }
            ++n_out;
        }
    }
    if (count) *count = n_out;
    return 0;
}

/* Thread-pooled batch of project_ calls. Threading shape of the reference: worker threads over
 * chunks of work (ref: include/impl/scene.hpp:146-166). `points_evaluated` sums `tried` per
 * hypothesis (= |subset| - saved). */
void tmo_score_hypotheses(const tmo_scene* s, const tmo_model* m, const float* poses16,
                          uint32_t n_hyp, const int32_t* subset, uint32_t n_subset,
                          float accept_prob, float dist_thres, int early_drop, int threads,
                          uint32_t* counts, double* scores, uint64_t* points_evaluated) {
    const uint32_t S = subset ? n_subset : s->n;
    std::atomic<uint64_t> evaluated(0);
    std::atomic<uint32_t> next(0);
    auto worker = [&]() {
        uint64_t local = 0;
        for (;;) {
            uint32_t h = next.fetch_add(1);
            if (h >= n_hyp) break;
            double sc = 0;
            uint32_t saved = 0;
            uint32_t c = tmo_project(s, m, subset, n_subset, poses16 + (size_t)h * 16, accept_prob,
                                     dist_thres, early_drop, nullptr, nullptr, &sc, &saved);
            if (counts) counts[h] = c;
            if (scores) scores[h] = sc;
            local += S - saved;
        }
        evaluated += local;
    };
    if (threads <= 1) {
        worker();
    } else {
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; ++t) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
    if (points_evaluated) *points_evaluated = evaluated.load();
}

/* ref: include/impl/scene.hpp:273 radius_search_inclusive(m.diameter(), p1): membership defined as
 * squared distance <= radius*radius (PCL/FLANN result order and boundary are unpinned); returned in
 * ascending index order. */
uint32_t tmo_scene_radius_subset(const tmo_scene* s, uint32_t center_idx, float radius, int32_t* out,
                                 uint32_t cap) {
    const float* c = s->cloud.data() + (size_t)center_idx * 12;
    float r2 = radius * radius;
    uint32_t n = 0;
    for (uint32_t i = 0; i < s->n; ++i) {
        const float* p = s->cloud.data() + (size_t)i * 12;
        float d2 = sqn(sub(s_pos(p), s_pos(c)));
        if (d2 <= r2) {
            if (n < cap && out) out[n] = (int32_t)i;
            ++n;
        }
    }
    return n;
}

}  // extern "C"

/* =============================================================================================
 * traits math (a15). state = [radius, threshold, origin(3), g2l(16 col-major), l2g(16)]
 * =========================================================================================== */
namespace {

struct TState {
    float radius, threshold;
    v3 origin;
    float g2l[16], l2g[16];
};
void load_state(const float* st, TState& t) {
    t.radius = st[0];
    t.threshold = st[1];
    t.origin = {st[2], st[3], st[4]};
    memcpy(t.g2l, st + 5, 64);
    memcpy(t.l2g, st + 21, 64);
}
void store_state(const TState& t, float* st) {
    st[0] = t.radius;
    st[1] = t.threshold;
    st[2] = t.origin.x;
    st[3] = t.origin.y;
    st[4] = t.origin.z;
    memcpy(st + 5, t.g2l, 64);
    memcpy(st + 21, t.l2g, 64);
}
/* g2l = [rows r0,r1,r2 | -R*origin]; l2g = rigid inverse written out (Eigen's general inverse()
 * is unpinnable; for an orthonormal frame the inverse is [R^T | origin]) */
void frame_from_rows(v3 r0, v3 r1, v3 r2, v3 origin, TState& t) {
    for (int i = 0; i < 16; ++i) t.g2l[i] = t.l2g[i] = 0.f;
    float R[3][3] = {{r0.x, r0.y, r0.z}, {r1.x, r1.y, r1.z}, {r2.x, r2.y, r2.z}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            t.g2l[c * 4 + r] = R[r][c];
            t.l2g[r * 4 + c] = R[r][c];
        }
    v3 no = {-origin.x, -origin.y, -origin.z};
    v3 tr = {dot(r0, no), dot(r1, no), dot(r2, no)};
    t.g2l[12] = tr.x;
    t.g2l[13] = tr.y;
    t.g2l[14] = tr.z;
    t.g2l[15] = 1.f;
    t.l2g[12] = origin.x;
    t.l2g[13] = origin.y;
    t.l2g[14] = origin.z;
    t.l2g[15] = 1.f;
    t.origin = origin;
}
/* Eigen unitOrthogonal() for 3-vectors (Eigen/src/Geometry/OrthoMethods.h), prec = 1e-5f */
v3 unit_orthogonal(v3 s) {
    auto much_smaller = [](float a, float b) { return fabsf(a) <= fabsf(b) * 1e-5f; };
    if (!much_smaller(s.x, s.z) || !much_smaller(s.y, s.z)) {
        float invnm = 1.f / sqrtf(s.x * s.x + s.y * s.y);
        return {-s.y * invnm, s.x * invnm, 0.f};
    }
    float invnm = 1.f / sqrtf(s.y * s.y + s.z * s.z);
    return {0.f, -s.z * invnm, s.y * invnm};
}
v3 g2l_point(const TState& t, v3 p) {
    float o[4];
    mat4_mul(t.g2l, p.x, p.y, p.z, 1.f, o);
    return {o[0], o[1], o[2]};
}
/* Jacobian of the cylinder unrolling applied to a local direction.
 * ref: include/impl/cylinder_traits.hpp:142-145 */
v3 cyl_jac(float r, v3 loc, v3 lt) {
    float n = sqrtf(loc.x * loc.x + loc.y * loc.y);
    float s = n * n;
    v3 t;
    t.x = lt.x * (-r * loc.y / s) + lt.y * r * loc.x / s;
    t.y = lt.z;
    t.z = lt.x * loc.x / n + lt.y * loc.y / n;
    return t;
}

}  // namespace

extern "C" {

int tmo_traits_init_from_samples(int kind, const float* model_state, const float* samples,
                                 float* state) {
    TState ms, t;
    load_state(model_state, ms);
    t.radius = 0.f;
    t.threshold = ms.threshold;
    if (kind == TMO_TRAITS_IDENTITY) { /* ref: identity_traits.hpp:23-29 */
        frame_from_rows({1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}, t);
        t.threshold = 0.f;
    } else if (kind == TMO_TRAITS_PLANE) { /* ref: plane_traits.hpp:44-61 */
        v3 n = s_nrm(samples);
        v3 r1 = unit_orthogonal(n);
        v3 r0 = normalized(cross(r1, n));
        frame_from_rows(r0, r1, n, s_pos(samples), t);
    } else if (kind == TMO_TRAITS_PLANE2) { /* ref: plane2_traits.hpp:48-84 */
        v3 s1 = s_pos(samples), s2 = s_pos(samples + 12), s3 = s_pos(samples + 24);
        v3 d1 = sub(s2, s1), d2 = sub(s3, s1);
        v3 nrm = normalized(cross(d1, d2));
        float c1 = dot(s_nrm(samples), nrm), c2 = dot(s_nrm(samples + 12), nrm),
              c3 = dot(s_nrm(samples + 24), nrm);
        if (1.f - fabsf(c1) > 0.1f) return 1;
        if (1.f - fabsf(c2) > 0.1f) return 1;
        if (1.f - fabsf(c3) > 0.1f) return 1;
        d1 = sub(d1, mul(nrm, dot(d1, nrm)));
        d1 = normalized(d1);
        d2 = normalized(cross(nrm, d1));
        frame_from_rows(d1, d2, nrm, s1, t);
    } else { /* cylinder, ref: cylinder_traits.hpp:53-97 */
        v3 p1 = s_pos(samples), p2 = s_pos(samples + 12);
        v3 n1 = s_nrm(samples), n2 = s_nrm(samples + 12);
        v3 w = sub(add(n1, p1), p2);
        float a = dot(n1, n1), b = dot(n1, n2), c = dot(n2, n2), d = dot(n1, w), e = dot(n2, w);
        float den = a * c - b * b;
        float sc, tc;
        if ((double)den < 1e-8) {
            sc = 0.f;
            tc = (b > c ? d / b : e / c);
        } else {
            sc = (b * e - c * d) / den;
            tc = (a * e - b * d) / den;
        }
        v3 origin = add(add(p1, n1), mul(n1, sc));
        v3 axis = normalized(sub(add(p2, mul(n2, tc)), origin));
        v3 ref = unit_orthogonal(axis);
        t.radius = norm(cross(axis, sub(origin, p1)));
        frame_from_rows(ref, cross(ref, axis), axis, origin, t);
    }
    store_state(t, state);
    return 0;
}

int tmo_traits_project(int kind, const float* state, const float xyz[3], float uvw[3]) {
    if (kind == TMO_TRAITS_IDENTITY) { /* ref: identity_traits.hpp:32-35 */
        uvw[0] = xyz[0];
        uvw[1] = xyz[1];
        uvw[2] = xyz[2];
        return 1;
    }
    TState t;
    load_state(state, t);
    v3 loc = g2l_point(t, {xyz[0], xyz[1], xyz[2]});
    if (kind == TMO_TRAITS_PLANE) { /* ref: plane_traits.hpp:64-72 */
        if (fabsf(loc.z) > t.threshold) return 0;
        uvw[0] = loc.x;
        uvw[1] = loc.y;
        uvw[2] = loc.z;
        return 1;
    }
    if (kind == TMO_TRAITS_PLANE2) { /* ref: plane2_traits.hpp:86-88: no threshold */
        uvw[0] = loc.x;
        uvw[1] = loc.y;
        uvw[2] = loc.z;
        return 1;
    }
    /* cylinder, ref: cylinder_traits.hpp:100-114 */
    float height = sqrtf(loc.x * loc.x + loc.y * loc.y) - t.radius;
    if (fabsf(height) > t.threshold) return 0;
    uvw[1] = loc.z;
    uvw[2] = height / t.radius;
    float u = tmo_atan2f(loc.y, loc.x);
    if (u < 0.f) u = (float)((double)u + 2.0 * M_PI);
    u *= t.radius;
    uvw[0] = u;
    return 1;
}

void tmo_traits_unproject(int kind, const float* state, const float uvw[3], float xyz[3]) {
    if (kind == TMO_TRAITS_IDENTITY) {
        xyz[0] = uvw[0];
        xyz[1] = uvw[1];
        xyz[2] = uvw[2];
        return;
    }
    TState t;
    load_state(state, t);
    float loc[4] = {uvw[0], uvw[1], uvw[2], 1.f};
    if (kind == TMO_TRAITS_CYLINDER) { /* ref: cylinder_traits.hpp:117-131 (libm cos/sin) */
        float r = (uvw[2] + 1.f) * t.radius;
        float ang = uvw[0] / t.radius;
        loc[0] = r * std::cos(ang);
        loc[1] = r * std::sin(ang);
        loc[2] = uvw[1];
    }
    float o[4];
    mat4_mul(t.l2g, loc[0], loc[1], loc[2], 1.f, o);
    xyz[0] = o[0];
    xyz[1] = o[1];
    xyz[2] = o[2];
}

void tmo_traits_tangent(int kind, const float* state, const float* surfel, float out[3]) {
    v3 tg = s_tgt(surfel);
    v3 r;
    if (kind == TMO_TRAITS_IDENTITY) { /* ref: identity_traits.hpp:44-47 */
        r = tg;
    } else {
        TState t;
        load_state(state, t);
        v3 lt = mat3_mul(t.g2l, tg);
        if (kind == TMO_TRAITS_PLANE) /* ref: plane_traits.hpp:81-85 (not normalised) */
            r = lt;
        else if (kind == TMO_TRAITS_PLANE2) /* ref: plane2_traits.hpp:96-102 */
            r = normalized(lt);
        else { /* ref: cylinder_traits.hpp:134-153 */
            v3 loc = g2l_point(t, s_pos(surfel));
            lt = normalized(lt);
            r = normalized(cyl_jac(t.radius, loc, lt));
        }
    }
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}

void tmo_traits_normal(int kind, const float* state, const float* surfel, float out[3]) {
    v3 nr = s_nrm(surfel);
    v3 r;
    if (kind == TMO_TRAITS_IDENTITY) { /* ref: identity_traits.hpp:50-53 */
        r = nr;
    } else {
        TState t;
        load_state(state, t);
        if (kind == TMO_TRAITS_PLANE || kind == TMO_TRAITS_PLANE2) {
            /* ref: plane_traits.hpp:88-92, plane2_traits.hpp:105-109 */
            r = normalized(mat3_mul(t.g2l, nr));
        } else { /* ref: cylinder_traits.hpp:156-178 */
            v3 g1 = s_tgt(surfel);
            v3 g2 = normalized(cross(g1, nr));
            v3 loc = g2l_point(t, s_pos(surfel));
            v3 l1 = normalized(mat3_mul(t.g2l, g1));
            v3 l2 = normalized(mat3_mul(t.g2l, g2));
            v3 t1 = cyl_jac(t.radius, loc, l1);
            v3 t2 = cyl_jac(t.radius, loc, l2);
            r = normalized(cross(t1, t2));
        }
    }
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}

float tmo_traits_intrinsic_distance(int kind, const float* state, const float p0[3],
                                    const float p1[3]) {
    v3 a = {p0[0], p0[1], p0[2]}, b = {p1[0], p1[1], p1[2]};
    if (kind == TMO_TRAITS_IDENTITY || kind == TMO_TRAITS_PLANE) return norm(sub(b, a));
    if (kind == TMO_TRAITS_PLANE2) { /* ref: plane2_traits.hpp:112-116: 2-D */
        float dx = b.x - a.x, dy = b.y - a.y;
        return sqrtf(dx * dx + dy * dy);
    }
    /* ref: cylinder_traits.hpp:181-189 */
    float radius = state[0];
    float d_u = fabsf(b.x - a.x), d_v = fabsf(b.y - a.y);
    d_u = std::min(d_u, (float)(2.0 * M_PI * (double)radius) - d_u);
    return sqrtf(d_u * d_u + d_v * d_v);
}

/* ref: opencl/cylinder.cl:1-25 (atan2pi(y,x) = atan2(y,x)/pi, computed with tmo_atan2f) */
void tmo_cl_uv_project(const float loc4[4], const float cyl2ncoord16[16], float out4[4]) {
    float nc[4];
    for (int r = 0; r < 4; ++r)
        nc[r] = cyl2ncoord16[r] * loc4[0] + cyl2ncoord16[4 + r] * loc4[1] +
                cyl2ncoord16[8 + r] * loc4[2] + cyl2ncoord16[12 + r] * loc4[3];
    float u = tmo_atan2f(nc[1], nc[0]) / kPiF;
    if (u < 0.f) u += 2.f;
    u /= 2.f;
    out4[0] = u;
    out4[1] = nc[2];
    out4[2] = sqrtf(nc[0] * nc[0] + nc[1] * nc[1]) - 1.0f;
    out4[3] = 1.f;
}

/* ref: opencl/icp.cl:1-53 */
void tmo_cl_icp_projection(const float* pnts4, int n, const float* image4, const int img_size[2],
                           const int img_margin[2], const float* mat_align, const float* mat_uvw,
                           const float* mat_proj, const float* mat_norm, float max_corr_dist,
                           float* out_positions4, int* model_indices, int* scene_indices) {
    auto mm = [](const float* v, const float* mat, float* o) {
        for (int r = 0; r < 4; ++r)
            o[r] = mat[r] * v[0] + mat[4 + r] * v[1] + mat[8 + r] * v[2] + mat[12 + r] * v[3];
    };
    for (int index = 0; index < n; ++index) {
        float loc[4], uv[4], tmp[4], uvn[4];
        mm(pnts4 + 4 * (size_t)index, mat_align, loc);
        tmo_cl_uv_project(loc, mat_proj, uv);
        mm(uv, mat_norm, tmp);
        mm(tmp, mat_uvw, uvn);
        float extx = (float)(img_size[0] - 2 * img_margin[0] - 1);
        float exty = (float)(img_size[1] - 2 * img_margin[1] - 1);
        int px = (int)(uvn[0] * extx) + img_margin[0];
        int py = (int)(uvn[1] * exty) + img_margin[1];
        if (py == img_size[1]) py = img_size[1] - 1;
        model_indices[index] = -1;
        scene_indices[index] = -1;
        float* op = out_positions4 + 4 * (size_t)index;
        op[0] = op[1] = op[2] = op[3] = 0.f;
        if (px >= 0 && px < img_size[0] && py >= 0 && py < img_size[1]) {
            int idx = py * img_size[0] + px;
            float dx = image4[4 * (size_t)idx] - uvn[0], dy = image4[4 * (size_t)idx + 1] - uvn[1];
            float dist = sqrtf(dx * dx + dy * dy);
            op[3] = dist;
            if (dist < max_corr_dist) {
                model_indices[index] = idx;
                scene_indices[index] = index;
                op[0] = uvn[0];
                op[1] = uvn[1];
                op[2] = uvn[2];
                op[3] = uvn[3];
            }
        }
    }
}

}  // extern "C"
