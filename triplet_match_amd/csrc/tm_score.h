/* tm_score.h — the (hypothesis, point) evaluation shared by the scoring kernels
 * (tm_scene.hip, tm_earlydrop.hip). Restates reference include/impl/scene.hpp:432-490. */
#ifndef TM_SCORE_H_
#define TM_SCORE_H_

#include <cmath>
#include <limits>

#include "tm_internal.h"
#include "tm_traits_math.h"

namespace tmx {

constexpr int kScorePPT = 4;               /* points per thread held in registers */
constexpr int kScoreTile = 256 * kScorePPT; /* points per workgroup */
constexpr double kFx = 1099511627776.0;     /* 2^40: fixed-point scale of the score sum */

/* Timing experiments (stage ablation, folded gathers) return WRONG results by design. They exist only in a
 * library built with -DTM_TIMING_EXPERIMENTS (python -m triplet_match_amd.build --timing-experiments writes
 * libtm_hip_experiments.so); in the product build the tests below are the constant false and the code behind them
 * is not compiled into the kernels. */
#ifdef TM_TIMING_EXPERIMENTS
#define TM_ABLATE_EQ(a, v) ((a).ablate == (v))
#define TM_ABLATE_GE(a, v) ((a).ablate >= (v))
/* ablate = 128 + bits: which memory stream of the item kernel is folded into a window that stays in L1 / L2 (wrong
 * results, timing only): 1 no orientation stage, 2 voxel entries, 4 model positions, 8 brick codes, 16 model refs,
 * 32 scene refs */
#define TM_ABLATE_BIT(a, b) ((a).ablate >= 128 && (((a).ablate - 128) & (b)) != 0)
#else
#define TM_ABLATE_EQ(a, v) false
#define TM_ABLATE_BIT(a, b) false
#define TM_ABLATE_GE(a, v) false
#endif

struct score_args {
    const float4* spos;
    const float* spos3; /* culled kernel: packed xyz, 12 B per point */
    const float4* sref;
    const float4* mpos;
    const float* mpos3; /* culled kernel: packed model xyz, 12 B per point; the is_tangent flag rides in mref.w */
    const float4* mref;
    const uint32_t* vox;
    const unsigned char* active; /* per original index, or null */
    const unsigned char* bcode;  /* brick distance codes of the per-point near filter, or null (filter off) */
    const unsigned char* bcode_tile; /* brick codes of the tile-level distance-field test (= bcode for 3-D L2; the
                                        all-points field under a projector's metric), or null */
    const uint32_t* super_near = nullptr; /* one bit per super-brick (16^3 voxels; index = brick >> 6): 1 = some brick
                                             of it passes the bcode test. null = no such filter. Early-drop kernels */
    int bx, by;                  /* brick grid dims (x, y) */
    uint32_t code_max;           /* reject a voxel hit when its brick code exceeds this */
    int ablate;                  /* read only under TM_TIMING_EXPERIMENTS (see TM_ABLATE_EQ); 0 = full kernel */
    int kind;                    /* traits whose intrinsic_distance applies (0 = 3-D L2) */
    float radius;                /* cylinder radius (wrap-around in u) */
    float dmax;                  /* thres itself, for the non-Euclidean metrics */
    const float* poses;
    uint32_t* counts;
    unsigned long long* scores_fx;
    uint4* stat_slots; /* one slot per workgroup (plain stores, no contention): evaluations, distance-passing,
                          in-grid, near; summed on the host only when timing is enabled */
    grid_t g;
    float d2max; /* largest f32 whose sqrtf is <= thres: d2 <= d2max <=> !(sqrtf(d2) > thres) */
    uint32_t n_hyp;
    uint32_t hyp_per_block;
    uint32_t n_super = 0; /* super-bricks behind super_near */
};

/* One (hypothesis, point) evaluation = reference include/impl/scene.hpp:432-490, in the stages between its
 * dependent loads. eval_point below is these stages in a row; kernels that keep several evaluations in flight
 * (k_ed_hist_wide, k_ed_prefix_group) call the stages of four points level by level, so the four loads of a level
 * are outstanding together. One source for the arithmetic either way. */
/* transform + in-grid test (:436-452, voxel_query's bounds): q = T p, brick and bricked voxel index */
__device__ __forceinline__ bool ev_grid(const score_args& a, const pose12& T, f3 p, f3& q, uint32_t& brick,
                                        uint32_t& lin) {
    q = xform_point(T, p);
    float vx = a.g.sx * q.x + a.g.tx;
    float vy = a.g.sy * q.y + a.g.ty;
    float vz = a.g.sz * q.z + a.g.tz;
    bool in = (vx > -1.f) & (vx < a.g.fex) & (vy > -1.f) & (vy < a.g.fey) & (vz > -1.f) & (vz < a.g.fez);
    brick = 0u;
    lin = 0u;
    if (!in) return false;
    const uint32_t ix = (uint32_t)(int)vx, iy = (uint32_t)(int)vy, iz = (uint32_t)(int)vz;
    brick = brick_of(ix, iy, iz, (uint32_t)a.bx, (uint32_t)a.by);
    lin = brick * 64u + (((iz & 3u) << 4) | ((iy & 3u) << 2) | (ix & 3u)); /* bricked layout */
    return true;
}
/* distance test of the voxel's model point (:462-466) */
__device__ __forceinline__ bool ev_dist(const score_args& a, f3 q, float4 mp) {
    if (a.kind == 0 || a.kind == 1) { /* identity / plane: 3-D L2 */
        float dx = q.x - mp.x, dy = q.y - mp.y, dz = q.z - mp.z;
        float d2 = (dx * dx + dy * dy) + dz * dz;
        return d2 <= a.d2max;
    }
    /* plane2 (2-D) / cylinder (wrap-around): the traits' intrinsic_distance */
    float dist = traits_distance(a.kind, a.radius, make3(q.x, q.y, q.z), make3(mp.x, mp.y, mp.z));
    return !(dist > a.dmax);
}
/* tangent flags (:472) and the score term (:476-486) in fixed point */
/* trunc(term * 2^40) for term >= 0 in integer arithmetic: the value is M * 2^(e - 150) with M = 2^23 | mantissa, so
 * the result is (M << 40) >> (150 - e) for biased exponents 87..150 and 0 below (term < 2^-40, denormals included).
 * Equal to (unsigned long long)(long long)((double)term * kFx) for every finite term < 2^24 (the fixed-point sum's own
 * packing gives out at 2^12); NaN / inf / >= 2^24 give 0. Eight full-rate integer instructions instead of eight f64
 * conversions (cvt, ldexp, trunc, ldexp, floor, fma, two cvt). */
__device__ __forceinline__ unsigned long long fx40_trunc(float term) {
    const uint32_t b = __float_as_uint(term) & 0x7FFFFFFFu;
    const uint32_t e = b >> 23;
    uint32_t hi = ((b & 0x7FFFFFu) | 0x800000u) << 8; /* M << 40 = hi : 0 */
    hi = (e - 87u <= 63u) ? hi : 0u;
    return ((unsigned long long)hi << 32) >> ((150u - e) & 63u);
}

__device__ __forceinline__ bool ev_term(const pose12& T, float4 sr, float4 mp, float4 mr, long long& fx) {
    if (sr.w != mp.w) return false; /* use_tangent != is_tangent, reference :472 */
    f3 r = xform_dir(T, make3(sr.x, sr.y, sr.z));
    float term = fabsf(dot3(r, make3(mr.x, mr.y, mr.z)));
    fx += (long long)fx40_trunc(term);
    return true;
}

/* Returns 1 if the point becomes a correspondence; accumulates the fixed-point score term. */
__device__ __forceinline__ uint32_t eval_point(const score_args& a, const pose12& T, f3 p,
                                               uint32_t sorted_idx, long long& fx, uint32_t& pass,
                                               uint32_t* model_idx, bool* in_grid = nullptr) {
    f3 q;
    uint32_t brick, lin;
    const bool in = ev_grid(a, T, p, q, brick, lin);
    if (in_grid) *in_grid = in;
    if (!in) return 0u;
    if (a.bcode) { /* exact-safe: every model point is farther than thres from this voxel's cell */
        if ((uint32_t)a.bcode[brick] > a.code_max) return 0u;
    }
    uint32_t m = a.vox[lin];
    float4 mp = a.mpos[m];
    if (!ev_dist(a, q, mp)) return 0u;
    ++pass;
    float4 sr = a.sref[sorted_idx];
    float4 mr = a.mref[m];
    if (!ev_term(T, sr, mp, mr, fx)) return 0u;
    if (model_idx) *model_idx = m;
    return 1u;
}

/* N evaluations level by level (see above): live[i] in = the point takes part; out: tail[i] (in grid), corr[i],
 * fx[i] (the score term). pose_at(i) = the pose of evaluation i (called twice: keep it cheap or a reference). */
template <int N, class PoseAt>
__device__ __forceinline__ void eval_points_staged(const score_args& a, PoseAt pose_at, const f3 (&p)[N],
                                                   const uint32_t (&sorted_idx)[N], const bool (&live)[N],
                                                   bool (&tail)[N], bool (&corr)[N], long long (&fx)[N]) {
    f3 q[N];
    uint32_t brick[N], lin[N], m[N];
    bool near[N];
    uint32_t code[N];
    float4 mp[N], sr[N], mr[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        tail[i] = live[i] && ev_grid(a, pose_at(i), p[i], q[i], brick[i], lin[i]);
        corr[i] = false;
        fx[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) code[i] = (tail[i] && a.bcode) ? (uint32_t)a.bcode[brick[i]] : 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) near[i] = tail[i] && code[i] <= a.code_max;
    /* three dependent levels, not five: the scene point's orientation record goes out together with the voxel entry
     * (its address never depended on anything), the model point's together with the model position (both hang on the
     * voxel entry); the distance test then gates arithmetic only. A near point that fails it has fetched two records for
     * nothing (a fifth of them on the bench batch); the grouped kernel gained 4 % from the same move. */
#pragma unroll
    for (int i = 0; i < N; ++i) {
        m[i] = near[i] ? a.vox[lin[i]] : 0u;
        sr[i] = near[i] ? a.sref[sorted_idx[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        mp[i] = near[i] ? a.mpos[m[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
        mr[i] = near[i] ? a.mref[m[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) near[i] = near[i] && ev_dist(a, q[i], mp[i]);
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (near[i]) corr[i] = ev_term(pose_at(i), sr[i], mp[i], mr[i], fx[i]);
}

/* Conservative culling test shared by the tile-culled kernels: can the sphere `tb` (centre xyz, radius w;
 * w < 0 = empty) of scene points, moved by T, reach the model's voxel box at all? rnorm = Frobenius norm
 * of the rotation block (bounds the stretch of a radius), abs_eps = absolute rounding slack. */
__device__ __forceinline__ bool sphere_hits_box(const score_args& a, const pose12& T, float rnorm,
                                                float abs_eps, float4 tb) {
    f3 c = xform_point(T, make3(tb.x, tb.y, tb.z));
    float rho = tb.w * rnorm * 1.0001f + abs_eps;
    float vx = a.g.sx * c.x + a.g.tx, vy = a.g.sy * c.y + a.g.ty, vz = a.g.sz * c.z + a.g.tz;
    float mx = a.g.sx * rho + 0.05f, my = a.g.sy * rho + 0.05f, mz = a.g.sz * rho + 0.05f;
    return (tb.w >= 0.f) & (vx + mx > -1.f) & (vx - mx < a.g.fex) & (vy + my > -1.f) & (vy - my < a.g.fey) &
           (vz + mz > -1.f) & (vz - mz < a.g.fez);
}

/* largest float t with sqrtf(t) <= thres (sqrtf is correctly rounded and monotone) */
static inline float sq_threshold(float thres) {
    if (!(thres >= 0.f)) return -1.f;
    float t = thres * thres;
    while (sqrtf(t) > thres) t = std::nextafter(t, 0.f);
    for (;;) {
        float u = std::nextafter(t, std::numeric_limits<float>::infinity());
        if (sqrtf(u) <= thres) t = u; else break;
    }
    return t;
}


}  // namespace tmx

#endif /* TM_SCORE_H_ */
