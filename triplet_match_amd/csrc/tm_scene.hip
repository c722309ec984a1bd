/*
 * tm_scene.hip — scene side of the hot path: hypothesis generation (a10, a12 inner body), batched
 * inlier scoring (a11/a14) and the best-record reduction. Replaces scene<Point>::impl
 * (reference include/impl/scene.hpp).
 *
 * HBM layout. Original order (addressed by sample index): opos[n] float4 (xyz, w = tangent_mask),
 * otan[n] float4 (tangent xyz, w = mask). Cell-sorted order (scoring): spos[n_pad] float4 (xyz,
 * w = bits of the original index; masked / padded / non-finite points carry NaN in x so that they
 * fail the in-grid test), sref[n_pad] float4 (tangent if tangent_mask else normal, w = tangent_mask),
 * tiles[n_pad/64] float4 (bounding sphere of 64 consecutive sorted points). Points are sorted along a
 * Morton curve so that a wave's 64 points are spatial neighbours: the voxel-hit branch is taken by
 * whole waves or not at all, and a hypothesis can skip every tile whose sphere misses the model's
 * voxel box.
 */
#include <cstring>

#include "tm_internal.h"
#include "tm_icp_math.h"
#include "tm_table.h"
#include "tm_score.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <limits>
#include <chrono>
#include <thread>

namespace tmx {

/* Brute force: every workgroup keeps a tile of scene points in registers and streams a chunk of
 * hypotheses (pose in SGPRs) over it. VALU-bound: ~30 ops per evaluation on the miss path. */
template <bool SUBSET>
__global__ __launch_bounds__(256) void k_score_brute(score_args a) {
    const uint32_t tile0 = blockIdx.x * kScoreTile;
    f3 p[kScorePPT];
#pragma unroll
    for (int k = 0; k < kScorePPT; ++k) {
        float4 v = a.spos[tile0 + k * 256 + threadIdx.x];
        if (SUBSET) {
            uint32_t orig = __float_as_uint(v.w);
            if (!(v.x == v.x) || !a.active[orig]) v.x = __builtin_nanf("");
        }
        p[k] = make3(v.x, v.y, v.z);
    }
    const uint32_t h0 = blockIdx.y * a.hyp_per_block;
    const uint32_t h1 = min(h0 + a.hyp_per_block, a.n_hyp);
    uint32_t pass_total = 0;
    for (uint32_t h = h0; h < h1; ++h) {
        const pose12 T = load_pose(a.poses + (size_t)h * 16);
        uint32_t cnt = 0, pass = 0;
        long long fx = 0;
#pragma unroll
        for (int k = 0; k < kScorePPT; ++k)
            cnt += eval_point(a, T, p[k], tile0 + k * 256 + threadIdx.x, fx, pass, nullptr);
        if (__ballot(pass != 0)) { /* wave-uniform: most (tile, hypothesis) pairs miss the grid */
            for (int off = 32; off > 0; off >>= 1) {
                cnt += __shfl_down(cnt, off);
                pass += __shfl_down(pass, off);
                fx += __shfl_down(fx, off);
            }
            if ((threadIdx.x & 63) == 0) {
                if (cnt) {
                    atomicAdd(&a.counts[h], cnt);
                    atomicAdd(&a.scores_fx[h], (unsigned long long)fx);
                }
                pass_total += pass;
            }
        }
    }
    __shared__ uint32_t s_pass[4];
    if ((threadIdx.x & 63) == 0) s_pass[threadIdx.x >> 6] = pass_total;
    __syncthreads();
    if (threadIdx.x == 0)
        a.stat_slots[blockIdx.y * gridDim.x + blockIdx.x] =
            make_uint4(0u, (s_pass[0] + s_pass[1]) + (s_pass[2] + s_pass[3]), 0u, 0u);
}

/* ------------------------------------------------------------------------------------------------
 * Culled scoring: one workgroup per hypothesis. Phase 1 tests every tile's bounding sphere against
 * the model's voxel box under this pose (conservative margins, so no in-grid point is ever skipped)
 * and compacts the surviving tile ids into LDS; phase 2 evaluates the surviving tiles, four tiles
 * per wave iteration with predicated (branch-free) gathers so that four independent
 * voxel -> model-point gather chains are in flight per lane. No atomics: one store per hypothesis.
 * ---------------------------------------------------------------------------------------------- */
/* >>> scoring kernels: bench.py ties the committed PMC passes (profiles/rNN/traffic.json) to a hash of the text between
 * this marker and its closing twin, plus tm_score.h / tm_math.h / tm_table.h — a change in here needs new passes
 * (tools/profile_round5.sh), a change elsewhere in this file does not */
constexpr int kCullChunk = 1024; /* tiles culled per round; each may leave up to 4 sub-tiles in the 16 KiB list */

struct cull_args {
    const float4* tiles;
    const float4* supers; /* bounding spheres of 64-tile groups */
    const float4* subs;   /* bounding spheres of the four 16-point quarters of every tile */
    uint32_t n_tiles;
    float coord_max; /* max |coordinate| of the scene, for the rounding margin */
    float inv_q;     /* 1 / brick code quantum */
    float tcode;     /* (thres + slack) / q + 1: a brick code above this is farther than thres */
    const uint32_t* centers; /* per hypothesis: scene (original) index of the ball centre, or null */
    const float4* opos;      /* scene positions in original order (ball centres) */
    float ball_r, ball_r2;   /* radius and radius*radius of the neighbourhood restriction */
    const unsigned char* active_sorted; /* SUBSET: per sorted slot, 1 = the point is in the subset */
};

__device__ __forceinline__ bool tile_hits(const score_args& a, const pose12& T, float rnorm, float abs_eps,
                                          float ca_tcode, float ca_inv_q, float4 tb) {
    f3 c = xform_point(T, make3(tb.x, tb.y, tb.z));
    float rho = tb.w * rnorm * 1.0001f + abs_eps;
    float vx = a.g.sx * c.x + a.g.tx, vy = a.g.sy * c.y + a.g.ty, vz = a.g.sz * c.z + a.g.tz;
    float mx = a.g.sx * rho + 0.05f, my = a.g.sy * rho + 0.05f, mz = a.g.sz * rho + 0.05f;
    bool hit = (tb.w >= 0.f) & (vx + mx > -1.f) & (vx - mx < a.g.fex) & (vy + my > -1.f) & (vy - my < a.g.fey) &
               (vz + mz > -1.f) & (vz - mz < a.g.fez);
    if (!hit) return false;
    /* distance-field test: the NN distance is 1-Lipschitz, so every point x of the tile has
     * NN(x) >= NN(c') - rho >= code(brick(c'))*q - sqrt(3)*cell - rho. If that exceeds thres no point of
     * the tile can become a correspondence (slack and the +1 code of safety are inside tcode). */
    if ((vx > -1.f) & (vx < a.g.fex) & (vy > -1.f) & (vy < a.g.fey) & (vz > -1.f) & (vz < a.g.fez)) {
        const uint32_t ix = (uint32_t)(int)vx, iy = (uint32_t)(int)vy, iz = (uint32_t)(int)vz;
        const float code = (float)a.bcode_tile[brick_of(ix, iy, iz, (uint32_t)a.bx, (uint32_t)a.by)];
        if (code > ca_tcode + rho * ca_inv_q * 1.0001f) return false;
    }
    return true;
}

/* tile_hits in two steps, so that a caller can put several brick-code loads in flight before it needs
 * the first: tile_probe does the arithmetic, tile_decide takes the loaded code. Same result. */
struct tile_probe_t {
    bool hit;       /* the sphere reaches the voxel box */
    bool need;      /* the distance-field test applies (centre inside the grid) */
    uint32_t brick; /* brick of the centre (0 when !need) */
    float lim;      /* reject when code > lim */
};
__device__ __forceinline__ tile_probe_t tile_probe(const score_args& a, const pose12& T, float rnorm, float abs_eps,
                                                   float ca_tcode, float ca_inv_q, float4 tb) {
    tile_probe_t r;
    f3 c = xform_point(T, make3(tb.x, tb.y, tb.z));
    float rho = tb.w * rnorm * 1.0001f + abs_eps;
    float vx = a.g.sx * c.x + a.g.tx, vy = a.g.sy * c.y + a.g.ty, vz = a.g.sz * c.z + a.g.tz;
    float mx = a.g.sx * rho + 0.05f, my = a.g.sy * rho + 0.05f, mz = a.g.sz * rho + 0.05f;
    r.hit = (tb.w >= 0.f) & (vx + mx > -1.f) & (vx - mx < a.g.fex) & (vy + my > -1.f) & (vy - my < a.g.fey) &
            (vz + mz > -1.f) & (vz - mz < a.g.fez);
    r.need = r.hit & (vx > -1.f) & (vx < a.g.fex) & (vy > -1.f) & (vy < a.g.fey) & (vz > -1.f) & (vz < a.g.fez);
    const uint32_t ix = r.need ? (uint32_t)(int)vx : 0u, iy = r.need ? (uint32_t)(int)vy : 0u,
                   iz = r.need ? (uint32_t)(int)vz : 0u;
    r.brick = brick_of(ix, iy, iz, (uint32_t)a.bx, (uint32_t)a.by);
    r.lim = ca_tcode + rho * ca_inv_q * 1.0001f;
    return r;
}
__device__ __forceinline__ bool tile_decide(const tile_probe_t& r, uint32_t code) {
    return r.hit & !(r.need & ((float)code > r.lim));
}

/* L2 = the model's metric is the 3-D Euclidean one (identity / plane): squared-distance comparison; otherwise the
 * traits' intrinsic_distance (plane2 / cylinder). NEAR = the distance-field tile test and the brick near-filter
 * apply: always for L2 (codes of the voxel build), for the other metrics with the codes tm_model_set_metric builds
 * under that metric (a.bcode / a.bcode_tile); NEAR = false leaves the sphere-vs-box test as the only culling.
 * SUBSET = only the points flagged in ca.active_sorted (sorted order) take part. */
template <bool L2, bool SUBSET, bool NEAR = L2>
__global__ __launch_bounds__(256, 5) void k_score_culled(score_args a, cull_args ca) {
    __shared__ uint32_t s_list[4 * kCullChunk]; /* surviving sub-tiles (16 points each) of the round */
    __shared__ uint32_t s_tiles[kCullChunk];    /* surviving tiles of the round */
    __shared__ uint32_t s_n, s_nt;
    __shared__ unsigned long long s_smask;
    __shared__ uint32_t s_cnt[4], s_pass[4];
    __shared__ unsigned long long s_fx[4];
    /* XCD-aware block -> hypothesis map (speed only): blocks b and b+8 share an XCD's L2, and
     * consecutive hypotheses come from the same scene sample pair, i.e. touch the same scene tiles and
     * voxels. Give each XCD a contiguous run of hypotheses so those re-reads hit its L2. Bijective. */
    uint32_t h;
    {
        const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
        const uint32_t q = a.n_hyp >> 3, r = a.n_hyp & 7u;
        h = x * q + min(x, r) + i;
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (TM_ABLATE_EQ(a, 6)) return;
    const pose12 T = load_pose(a.poses + (size_t)h * 16);
    const float rnorm = sqrtf(((T.r00 * T.r00 + T.r10 * T.r10) + (T.r20 * T.r20 + T.r01 * T.r01)) +
                              ((T.r11 * T.r11 + T.r21 * T.r21) + (T.r02 * T.r02 + T.r12 * T.r12)) +
                              T.r22 * T.r22);
    const float abs_eps = 1e-6f * (3.f * rnorm * ca.coord_max + fabsf(T.tx) + fabsf(T.ty) + fabsf(T.tz) + 1.f);
    uint32_t cnt = 0, pass = 0, n_in = 0, n_near = 0;
    long long fx = 0;
    unsigned long long tiles_done = 0;
    /* optional neighbourhood restriction: ball around the hypothesis' first sample point */
    const bool ball = ca.centers != nullptr;
    f3 bc = make3(0.f, 0.f, 0.f);
    if (ball) {
        float4 C = ca.opos[ca.centers[h]];
        bc = make3(C.x, C.y, C.z);
    }
    const float ball_margin = 1e-5f * ca.ball_r + 4e-6f * ca.coord_max + 1e-12f;
    /* a hypothesis may be split over gridDim.y workgroups that take the culling rounds in turn: the counts
     * and fixed-point sums are integers, so the parts simply add up (atomics, order-free) */
    for (uint32_t t0 = blockIdx.y * kCullChunk; t0 < ca.n_tiles; t0 += kCullChunk * gridDim.y) {
        /* level 1: the chunk's 64 super-tiles (64 tiles each), one lane each, wave 0 */
        if (wave == 0) {
            const uint32_t u = (t0 >> 6) + lane;
            bool hit = false;
            if (lane < (uint32_t)(kCullChunk / 64) && u * 64u < ca.n_tiles) {
                float4 sb = ca.supers[u];
                hit = sphere_hits_box(a, T, rnorm, abs_eps, sb);
                if (ball) hit = hit && !(len(make3(sb.x, sb.y, sb.z) - bc) > ca.ball_r + sb.w * 1.00001f + ball_margin);
            }
            unsigned long long mask = __ballot(hit);
            if (lane == 0) {
                s_smask = mask;
                s_n = 0;
                s_nt = 0;
            }
        }
        __syncthreads();
        const unsigned long long smask = s_smask;
        /* level 2: wave w owns super-tiles w, w+4, ... of the chunk: wave-uniform skip */
        for (uint32_t r = wave; r < (uint32_t)(kCullChunk / 64); r += 4) {
            if (TM_ABLATE_EQ(a, 5)) break;
            if (!((smask >> r) & 1ull)) continue;
            const uint32_t t = t0 + r * 64u + lane;
            bool hit = false;
            if (t < ca.n_tiles) {
                float4 tb = ca.tiles[t];
                hit = NEAR ? tile_hits(a, T, rnorm, abs_eps, ca.tcode, ca.inv_q, tb)
                           : sphere_hits_box(a, T, rnorm, abs_eps, tb);
                if (ball) hit = hit && !(len(make3(tb.x, tb.y, tb.z) - bc) > ca.ball_r + tb.w * 1.00001f + ball_margin);
            }
            if (hit) s_tiles[atomicAdd(&s_nt, 1u)] = t;
        }
        __syncthreads();
        /* level 3: the four 16-point quarters of every surviving tile, the same tests with their own
         * (smaller) spheres; one (tile, quarter) per thread */
        const uint32_t n_quarters = 4u * s_nt;
        for (uint32_t idx = threadIdx.x; idx < n_quarters; idx += 256) {
            const uint32_t sub = s_tiles[idx >> 2] * 4u + (idx & 3u);
            const float4 sb = ca.subs[sub];
            bool hit = NEAR ? tile_hits(a, T, rnorm, abs_eps, ca.tcode, ca.inv_q, sb)
                            : sphere_hits_box(a, T, rnorm, abs_eps, sb);
            if (ball) hit = hit && !(len(make3(sb.x, sb.y, sb.z) - bc) > ca.ball_r + sb.w * 1.00001f + ball_margin);
            if (hit) s_list[atomicAdd(&s_n, 1u)] = sub;
        }
        __syncthreads();
        const uint32_t n = TM_ABLATE_EQ(a, 4) ? 0u : s_n;
        tiles_done += n;
        /* a wave's 64 lanes take four sub-tiles at a time (lane >> 4 picks the sub-tile), four such groups
         * in flight: 16 list entries per iteration */
        for (uint32_t i = wave * 16; i < n; i += 64) {
            /* stage 1: load 4 tiles' points, then transform + in-grid, then 4 brick-code loads */
            f3 q[4];
            uint32_t sidx[4], lin[4], m[4], brick[4];
            bool near[4], in[4];
            float4 P[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t e = i + 4u * k + (lane >> 4);
                const bool have = e < n;
                sidx[k] = (have ? s_list[e] : s_list[i]) * 16u + (lane & 15u);
                const float* sp = a.spos3 + 3u * (size_t)sidx[k]; /* 12 B per point: 12 lines per tile, not 16 */
                if (TM_ABLATE_GE(a, 8)) { /* timing experiment: no per-point loads (one broadcast sphere per 16 lanes) */
                    const float4 sb = ca.subs[sidx[k] >> 4];
                    P[k] = make_float4(sb.x + 1e-4f * (float)(lane & 3u), sb.y + 1e-4f * (float)((lane >> 2) & 3u), sb.z, 0.f);
                } else
                P[k] = make_float4(sp[0], sp[1], sp[2], 0.f);
                in[k] = have;
                if (SUBSET) in[k] = in[k] & (ca.active_sorted[sidx[k]] != 0);
                if (ball) { /* exact membership: (dx*dx + dy*dy) + dz*dz <= r*r on the untransformed point */
                    f3 d = make3(P[k].x, P[k].y, P[k].z) - bc;
                    in[k] = in[k] & (len2(d) <= ca.ball_r2);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q[k] = xform_point(T, make3(P[k].x, P[k].y, P[k].z));
                float vx = a.g.sx * q[k].x + a.g.tx, vy = a.g.sy * q[k].y + a.g.ty, vz = a.g.sz * q[k].z + a.g.tz;
                in[k] = in[k] & (vx > -1.f) & (vx < a.g.fex) & (vy > -1.f) & (vy < a.g.fey) & (vz > -1.f) &
                        (vz < a.g.fez);
                uint32_t ix = in[k] ? (uint32_t)(int)vx : 0u, iy = in[k] ? (uint32_t)(int)vy : 0u,
                         iz = in[k] ? (uint32_t)(int)vz : 0u;
                brick[k] = brick_of(ix, iy, iz, (uint32_t)a.bx, (uint32_t)a.by);
                lin[k] = brick[k] * 64u + (((iz & 3u) << 4) | ((iy & 3u) << 2) | (ix & 3u));
            }
            uint32_t code[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) code[k] = NEAR ? (uint32_t)a.bcode[brick[k]] : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                near[k] = in[k] & (code[k] <= a.code_max);
                n_in += in[k] ? 1u : 0u;
                n_near += near[k] ? 1u : 0u;
                lin[k] = near[k] ? lin[k] : 0u;
                if (TM_ABLATE_EQ(a, 7) || TM_ABLATE_EQ(a, 9)) lin[k] &= 0x3FFFFu; /* timing experiment: voxel gathers stay in 1 MB */
            }
            if (TM_ABLATE_EQ(a, 3)) { cnt += near[0] + near[1] + near[2] + near[3]; continue; }
            /* stage 2: voxel gathers */
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = a.vox[lin[k]];
            if (TM_ABLATE_EQ(a, 2)) { cnt += m[0] + m[1] + m[2] + m[3]; continue; }
            /* stage 3: model point gathers + distance test */
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* mp = a.mpos3 + 3u * (size_t)m[k]; /* packed xyz: 5.3 points per 64-B line */
                if (L2) {
                    float dx = q[k].x - mp[0], dy = q[k].y - mp[1], dz = q[k].z - mp[2];
                    float d2 = (dx * dx + dy * dy) + dz * dz;
                    ok[k] = near[k] & (d2 <= a.d2max);
                } else { /* reference voxel_distance: !(intrinsic_distance > thres), on the square (see below) */
                    float d2 = traits_distance2(a.kind, a.radius, q[k], make3(mp[0], mp[1], mp[2]));
                    ok[k] = near[k] & !(d2 > a.d2max);
                }
            }
            if (TM_ABLATE_EQ(a, 1)) { cnt += ok[0] + ok[1] + ok[2] + ok[3]; continue; }
            /* stage 4: orientation term for the distance-passing points; the eight loads are issued before
             * the first use so that they overlap */
            if (__ballot(ok[0] | ok[1] | ok[2] | ok[3])) {
#pragma unroll
                for (int k0 = 0; k0 < 4; k0 += 2) {
                    float4 sr[2], mr[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (ok[k0 + j]) {
                            sr[j] = a.sref[sidx[k0 + j]];
                            mr[j] = a.mref[m[k0 + j]]; /* w = is_tangent of the model point */
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (ok[k0 + j]) {
                            ++pass;
                            if (sr[j].w == mr[j].w) { /* use_tangent == is_tangent, reference :472 */
                                f3 r = xform_dir(T, make3(sr[j].x, sr[j].y, sr[j].z));
                                float term = fabsf(dot3(r, make3(mr[j].x, mr[j].y, mr[j].z)));
                                fx += (long long)fx40_trunc(term);
                                ++cnt;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off);
        pass += __shfl_down(pass, off);
        fx += __shfl_down(fx, off);
        n_in += __shfl_down(n_in, off);
        n_near += __shfl_down(n_near, off);
    }
    __shared__ uint32_t s_in[4], s_near[4];
    if (lane == 0) {
        s_in[wave] = n_in;
        s_near[wave] = n_near;
        s_cnt[wave] = cnt;
        s_pass[wave] = pass;
        s_fx[wave] = (unsigned long long)fx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c_all = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        const unsigned long long f_all = (s_fx[0] + s_fx[1]) + (s_fx[2] + s_fx[3]);
        const uint4 st = make_uint4((uint32_t)(tiles_done * 16ull), (s_pass[0] + s_pass[1]) + (s_pass[2] + s_pass[3]),
                                    (s_in[0] + s_in[1]) + (s_in[2] + s_in[3]),
                                    (s_near[0] + s_near[1]) + (s_near[2] + s_near[3]));
        if (gridDim.y == 1) {
            a.counts[h] = c_all;
            a.scores_fx[h] = f_all;
            a.stat_slots[h] = st;
        } else { /* outputs were zeroed by the host */
            if (c_all) atomicAdd(&a.counts[h], c_all);
            if (f_all) atomicAdd(&a.scores_fx[h], f_all);
            uint32_t* sp = reinterpret_cast<uint32_t*>(&a.stat_slots[h]);
            if (st.x) atomicAdd(sp, st.x);
            if (st.y) atomicAdd(sp + 1, st.y);
            if (st.z) atomicAdd(sp + 2, st.z);
            if (st.w) atomicAdd(sp + 3, st.w);
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Grouped scoring: the hypotheses of one scene sample pair all map the first sample onto a model point,
 * so they reach the same neighbourhood of the scene (the reference's `nn`, include/impl/scene.hpp:273).
 * Instead of letting every hypothesis cull the whole scene and load its points for itself, groups of
 * 16 consecutive hypotheses share the work:
 *   k_group_cull   one thread per (group, 512-point chunk): which of the group's hypotheses can the
 *                  chunk's bounding sphere reach at all -> 16-bit mask per (group, chunk)
 *   rocprim select the (group, chunk) slots with a non-zero mask -> work items
 *   k_score_items  persistent waves; an item = one chunk staged ONCE into the wave's private LDS
 *                  (coalesced 16-B loads) and evaluated for every hypothesis of the mask: quarter-tile
 *                  test by a ballot, surviving quarters compacted, then the same four-chain evaluation
 *                  as k_score_culled with the points read from LDS.
 * Everything skipped is skipped by the same conservative sphere tests as before, the sums are integers,
 * so counts and fixed-point scores are identical to k_score_culled / k_score_brute.
 * ---------------------------------------------------------------------------------------------- */
typedef uint32_t group_mask_t; /* up to 32 hypotheses per group */
/* LDS of one wave of k_score_items<G>: chunk xyz + poses (3 float4 each) + (rnorm, abs_eps) + entries + sums */
constexpr int item_lds(int G, bool ball = false) { return 6144 + G * 48 + G * 8 + G * 32 * 2 + G * 8 + (ball ? G * 16 : 0); }
constexpr int kChunkPts = 512; /* 8 tiles = 32 quarter-tiles, 6 KB of packed xyz */
static_assert(kScoreTile % kChunkPts == 0, "the padded scene (a multiple of kScoreTile) must be whole chunks");
constexpr int kItemWaves = 4; /* waves per SIMD the item kernel's register budget is set for (127 VGPRs). Five
                                 waves = 96 VGPRs spill and lose: four chains 100 bytes per lane, 18.3 ms against
                                 14.8; three chains 72 bytes, 16.8 ms against 12.9 */

struct group_args {
    const float4* chunks;   /* bounding sphere per 512 sorted points */
    uint32_t n_chunks;
    float inv_chunks;       /* 1 / n_chunks (split_slot) */
    uint32_t n_groups;
    uint32_t group_hyps;    /* hypotheses per group: 16 or 32 */
    group_mask_t* masks;    /* [group][chunk] */
    const uint32_t* items;  /* slots with a non-zero mask */
    const uint32_t* n_items;
    const float2* aux;      /* per hypothesis: (rnorm, abs_eps) of the sphere tests */
    uint32_t* heads;        /* per XCD (128 bytes apart): the next item of its runs */
};

/* slot = group * n_chunks + chunk -> (group, chunk) without an integer division (~30 VALU instructions in a kernel that
 * is bound by VALU issue): a float estimate of the quotient, off by one at most for slots below 2^31 / 8, put right by
 * the remainder. */
__device__ __forceinline__ void split_slot(uint32_t slot, uint32_t n_chunks, float inv_chunks, uint32_t& group, uint32_t& chunk) {
    uint32_t g = (uint32_t)((float)slot * inv_chunks);
    uint32_t r = slot - g * n_chunks; /* wraps when g is one too large */
    if ((int32_t)r < 0) {
        --g;
        r += n_chunks;
    } else if (r >= n_chunks) {
        ++g;
        r -= n_chunks;
    }
    group = g;
    chunk = r;
}

struct nonzero_mask {
    __host__ __device__ bool operator()(group_mask_t m) const { return m != 0; }
};

/* per hypothesis: Frobenius norm of the rotation block and the absolute rounding slack of the sphere tests */
__device__ __forceinline__ float2 pose_aux(const pose12& T, float coord_max) {
    const float rnorm = sqrtf(((T.r00 * T.r00 + T.r10 * T.r10) + (T.r20 * T.r20 + T.r01 * T.r01)) +
                              ((T.r11 * T.r11 + T.r21 * T.r21) + (T.r02 * T.r02 + T.r12 * T.r12)) +
                              T.r22 * T.r22);
    return make_float2(rnorm, 1e-6f * (3.f * rnorm * coord_max + fabsf(T.tx) + fabsf(T.ty) + fabsf(T.tz) + 1.f));
}
__global__ __launch_bounds__(256) void k_pose_aux(const float* __restrict__ poses, uint32_t n_hyp, float coord_max,
                                                  float2* __restrict__ aux) {
    const uint32_t h = blockIdx.x * 256 + threadIdx.x;
    if (h >= n_hyp) return;
    aux[h] = pose_aux(load_pose(poses + (size_t)h * 16), coord_max);
}

/* One workgroup per (group, 256 super-tiles of 8 chunks). Phase 1: a thread = a super-tile, its sphere under each of the
 * group's poses (uniform: scalar loads); the (super-tile, pose) pairs that pass — one in ten — go to a list in LDS.
 * Phase 2: the list's pairs x 8 chunks as dense tasks (pose per lane from LDS), hits OR-ed into the masks in LDS.
 * With a thread per (group, chunk), or the chunk tests inside phase 1's loop, nearly every pose had some lane of the
 * wave to serve and the wave ran all 16 x 8 chunk tests mostly idle (0.26 ms per batch, VALU-bound). Same tests, hence
 * the same masks. */
constexpr uint32_t kCullSupers = 256, kCullMaxHyps = 32;
__global__ __launch_bounds__(256) void k_group_cull(score_args a, cull_args ca, group_args ga) {
    __shared__ float s_pose[kCullMaxHyps * 16];
    __shared__ float2 s_aux[kCullMaxHyps];
    __shared__ float4 s_ctr[kCullMaxHyps];
    __shared__ uint32_t s_m[kCullSupers * 8];
    __shared__ unsigned short s_pair[kCullSupers * kCullMaxHyps];
    __shared__ uint32_t s_np;
    const uint32_t group = blockIdx.x; /* groups along x: any batch size fits one launch */
    const uint32_t sup = blockIdx.y * kCullSupers + threadIdx.x, lane = threadIdx.x & 63u;
    const uint32_t h0 = group * ga.group_hyps;
    const uint32_t n_g = min(ga.group_hyps, a.n_hyp > h0 ? a.n_hyp - h0 : 0u);
    for (uint32_t i = threadIdx.x; i < n_g * 16u; i += 256u) s_pose[i] = a.poses[(size_t)h0 * 16 + i];
    if (threadIdx.x < n_g) {
        s_aux[threadIdx.x] = ga.aux[h0 + threadIdx.x];
        s_ctr[threadIdx.x] = ca.centers ? ca.opos[ca.centers[h0 + threadIdx.x]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (uint32_t i = threadIdx.x; i < kCullSupers * 8u; i += 256u) s_m[i] = 0u;
    if (threadIdx.x == 0) s_np = 0u;
    __syncthreads();
    const float4 ss = sup * 8u < ga.n_chunks ? ca.supers[sup] : make_float4(0.f, 0.f, 0.f, -1.f);
    for (uint32_t g = 0; g < n_g; ++g) {
        const pose12 T = load_pose(a.poses + (size_t)(h0 + g) * 16); /* uniform: the pose comes through scalar loads */
        const float2 ax = ga.aux[h0 + g];
        const bool hit = sphere_hits_box(a, T, ax.x, ax.y, ss);
        const unsigned long long b = __ballot(hit);
        if (b) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_np, (uint32_t)__popcll(b));
            base = (uint32_t)__shfl((int)base, 0);
            if (hit) s_pair[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)((threadIdx.x << 5) | g);
        }
    }
    __syncthreads();
    const uint32_t n_task = s_np * 8u;
    const float ball_margin = 1e-5f * ca.ball_r + 4e-6f * ca.coord_max + 1e-12f;
    for (uint32_t t = threadIdx.x; t < n_task; t += 256u) {
        const uint32_t pr = s_pair[t >> 3], k = t & 7u, sl = pr >> 5, g = pr & 31u;
        const uint32_t chunk = (blockIdx.y * kCullSupers + sl) * 8u + k;
        if (chunk >= ga.n_chunks) continue;
        const float4 sb = ga.chunks[chunk];
        const pose12 T = load_pose(s_pose + g * 16u);
        const float2 ax = s_aux[g];
        bool hit = sphere_hits_box(a, T, ax.x, ax.y, sb);
        if (ca.centers) { /* neighbourhood restriction: the chunk must reach the ball of the hypothesis' first point */
            const float4 C = s_ctr[g];
            hit = hit && !(len(make3(sb.x, sb.y, sb.z) - make3(C.x, C.y, C.z)) > ca.ball_r + sb.w * 1.00001f + ball_margin);
        }
        if (hit) atomicOr(&s_m[sl * 8u + k], 1u << g);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kCullSupers * 8u; i += 256u) {
        const uint32_t chunk = blockIdx.y * kCullSupers * 8u + i;
        if (chunk < ga.n_chunks) ga.masks[(size_t)group * ga.n_chunks + chunk] = (group_mask_t)s_m[i];
    }
}

/* Items in the order of the MODEL region they land in (option item_sort). L2 (4 MB per XCD) turns over every few
 * microseconds under the voxel gathers of ~500 concurrently running items per XCD that are scattered over the whole
 * near-surface shell of the table (~40 MB): a voxel line is fetched from the fabric ~1000 times per launch. The items
 * are therefore binned by the 16^3-voxel cell (Morton order of the cells) their chunk's centre lands in under the
 * item's first hypothesis (histogram, scan, scatter: three small launches, no host synchronisation, the item count
 * stays on the device), and the hand-out deals consecutive runs of the binned list to the XCDs: what runs on one XCD
 * at any time then gathers from a few neighbouring cells. The scene side loses nothing: a chunk is staged once per
 * item either way. Measured on the configs[1] batch: L2->fabric lines 4.81e8 -> 2.78e8 per launch, L2 hit rate
 * 33 % -> 60 %, kernel 9.04 -> 8.55 ms: the fabric traffic was not what bounds the kernel. */
__device__ __forceinline__ uint32_t spread3(uint32_t v) { /* 10 bits -> every third bit */
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(256) void k_item_bins(score_args a, group_args ga, uint32_t n_bins, uint32_t cell_shift,
                                                   uint32_t* __restrict__ bin_of, uint32_t* __restrict__ hist) {
    const uint32_t n_items = *ga.n_items;
    for (uint32_t it = blockIdx.x * 256 + threadIdx.x; it < n_items; it += gridDim.x * 256) {
        const uint32_t slot = ga.items[it];
        uint32_t group, chunk;
        split_slot(slot, ga.n_chunks, ga.inv_chunks, group, chunk);
        const uint32_t m = (uint32_t)ga.masks[slot];
        const uint32_t h = min(group * ga.group_hyps + (uint32_t)__builtin_ctz(m | 0x80000000u), a.n_hyp - 1u);
        const pose12 T = load_pose(a.poses + (size_t)h * 16);
        const float4 sb = ga.chunks[chunk];
        const f3 q = xform_point(T, make3(sb.x, sb.y, sb.z));
        const float vx = fminf(fmaxf(a.g.sx * q.x + a.g.tx, 0.f), a.g.fex - 1.f);
        const float vy = fminf(fmaxf(a.g.sy * q.y + a.g.ty, 0.f), a.g.fey - 1.f);
        const float vz = fminf(fmaxf(a.g.sz * q.z + a.g.tz, 0.f), a.g.fez - 1.f);
        uint32_t bin = 0;
        if (vx == vx && vy == vy && vz == vz) /* a NaN pose lands in bin 0; Morton order of the cells */
            bin = spread3((uint32_t)vx >> cell_shift) | (spread3((uint32_t)vy >> cell_shift) << 1) |
                  (spread3((uint32_t)vz >> cell_shift) << 2);
        bin = min(bin, n_bins - 1u);
        bin_of[it] = bin;
        /* consecutive items are neighbouring chunks of one group: a wave holds a handful of distinct bins, and one
         * atomic per (wave, bin) instead of one per item (776k per batch, the hot bins serialising: 0.24 ms) */
        unsigned long long todo = __ballot(true);
        const uint32_t lane = threadIdx.x & 63u;
        while (todo) {
            const uint32_t b = (uint32_t)__shfl((int)bin, __builtin_ctzll(todo));
            const unsigned long long same = __ballot(bin == b) & todo;
            if (lane == (uint32_t)__builtin_ctzll(same)) atomicAdd(&hist[b], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }
}

__global__ __launch_bounds__(256) void k_item_scatter(group_args ga, const uint32_t* __restrict__ bin_of,
                                                      uint32_t* __restrict__ cursor /* exclusive offsets, advanced here */,
                                                      uint32_t* __restrict__ sorted) {
    const uint32_t n_items = *ga.n_items;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t it = blockIdx.x * 256 + threadIdx.x; it < n_items; it += gridDim.x * 256) {
        const uint32_t bin = bin_of[it];
        /* one cursor atomic per (wave, bin), all of a wave's in one instruction (a returning atomic per distinct bin,
         * one after the other, is a microsecond each); the items of a wave keep their order inside the bin */
        unsigned long long todo = __ballot(true);
        uint32_t leader = lane, rank = 0, cnt = 1;
        while (todo) {
            const uint32_t b = (uint32_t)__shfl((int)bin, __builtin_ctzll(todo));
            const unsigned long long same = __ballot(bin == b) & todo;
            if (bin == b) {
                leader = (uint32_t)__builtin_ctzll(same);
                rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                cnt = (uint32_t)__popcll(same);
            }
            todo &= ~same;
        }
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&cursor[bin], cnt);
        base = (uint32_t)__shfl((int)base, (int)leader);
        sorted[base + rank] = ga.items[it];
    }
}

/* pose as staged in LDS: three float4 laid out for two-wide arithmetic on (x, y):
 *   A = (r00, r10, r01, r11)   B = (r02, r12, tx, ty)   C = (r20, r21, r22, tz)
 * so that T*p = ((A.xy * px + A.zw * py) + B.xy * pz) + B.zw for (x, y) — register pairs straight from the
 * 16-byte reads, v_pk_mul_f32 / v_pk_add_f32 without operand shuffling — and the dot of C with (p, 1) for z.
 * Same operations in the same association as xform_point / xform_dir (tm_math.h). */
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pose12 lds_pose(const float4* p) {
    const float4 A = p[0], B = p[1], C = p[2];
    pose12 T;
    T.r00 = A.x; T.r10 = A.y; T.r01 = A.z; T.r11 = A.w;
    T.r02 = B.x; T.r12 = B.y; T.tx = B.z; T.ty = B.w;
    T.r20 = C.x; T.r21 = C.y; T.r22 = C.z; T.tz = C.w;
    return T;
}
__device__ __forceinline__ f3 lds_xform_point(const float4 A, const float4 B, const float4 C, float px, float py, float pz) {
    const f2v a01 = {A.x, A.y}, a23 = {A.z, A.w}, b01 = {B.x, B.y}, b23 = {B.z, B.w};
    const f2v xy = ((a01 * px + a23 * py) + b01 * pz) + b23;
    return make3(xy.x, xy.y, ((C.x * px + C.y * py) + C.z * pz) + C.w);
}
__device__ __forceinline__ f3 lds_xform_dir(const float4 A, const float4 B, const float4 C, float px, float py, float pz) {
    const f2v a01 = {A.x, A.y}, a23 = {A.z, A.w}, b01 = {B.x, B.y};
    const f2v xy = (a01 * px + a23 * py) + b01 * pz;
    return make3(xy.x, xy.y, (C.x * px + C.y * py) + C.z * pz);
}

/* A gather as SGPR base + 32-bit byte offset (the `saddr` form of global_load): the 64-bit form costs three to four
 * VALU instructions of address arithmetic and a register pair per gather, four gathers per chain. Valid while the
 * table is below 4 GiB (k_score_items<.., W32>: voxel table, model and scene arrays, checked at the launch). */
template <class T>
__device__ __forceinline__ T ld_off32(const T* base, uint32_t byte_off) {
    asm volatile("" : "+v"(byte_off)); /* keeps the offset a 32-bit value: the combiner otherwise widens index << 2 to 64 bits */
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
struct f3pack {
    float x, y, z;
};

/* BALL: every hypothesis scores only the scene points within ca.ball_r of its own centre point (the reference's
 * nn = radius_search_inclusive(diameter, p1)): quarter-tiles conservatively by a sphere-sphere test, points exactly
 * by (dx*dx + dy*dy) + dz*dz <= r*r on the untransformed point, as k_score_culled does */
/* STATS = false: without the per-stage bookkeeping (evaluations, in-grid, near, passing: three or four VALU
 * instructions per chain) that only a measurement window with counters reads (tm_ctx_enable_timing(ctx, 1)) */
template <bool L2, int NC, int G, bool BALL = false, bool NEAR = L2, bool W32 = false, bool STATS = true>
__global__ __launch_bounds__(256, kItemWaves) void k_score_items(score_args a, cull_args ca, group_args ga) {
    /* 8 KiB of LDS per wave, private to it: no workgroup barrier inside the loop */
    __shared__ __align__(16) unsigned char s_raw[4][item_lds(G, BALL)];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n_items = *ga.n_items;
    /* XCD-aware split (speed only): blocks b and b + 8 share an L2; consecutive items belong to the same
     * group of hypotheses, whose voxel neighbourhoods overlap: give each XCD a contiguous range of items */
    const uint32_t xcd = blockIdx.x & 7u;
    uint32_t evals = 0, pass = 0, n_in = 0, n_near = 0;
    float* pts = reinterpret_cast<float*>(s_raw[wave]);                                   /* 512 x packed xyz */
    float4* poses = reinterpret_cast<float4*>(s_raw[wave] + 6144);                        /* 3 float4 per pose */
    float2* auxl = reinterpret_cast<float2*>(s_raw[wave] + 6144 + G * 48);       /* (rnorm, abs_eps) */
    unsigned short* ent = reinterpret_cast<unsigned short*>(s_raw[wave] + 6144 + G * 56); /* entries */
    unsigned long long* acc =
        reinterpret_cast<unsigned long long*>(s_raw[wave] + 6144 + G * 56 + G * 64); /* count << 52 | sum */
    float4* cen = reinterpret_cast<float4*>(s_raw[wave] + 6144 + G * 56 + G * 64 + G * 8); /* BALL: centre per hypothesis */
    const float ball_margin = 1e-5f * ca.ball_r + 4e-6f * ca.coord_max + 1e-12f;
    if (lane < (uint32_t)G) acc[lane] = 0ull;
    /* Items differ a lot in cost (1 to 16 hypotheses, 0 to 32 quarter-tiles each): with a static share per wave
     * the average wave was resident for 72 % of the kernel only (12.97 ms). Each wave therefore pulls its next
     * item from a counter of its XCD (one returning atomic per ~50 us item; 11.3 ms), and when the XCD's items
     * are exhausted it continues with the next XCD's (9.65 ms). The items of an XCD are runs of 2^run_shift
     * consecutive items handed out in order, so the waves of one L2 work on neighbouring items at any time
     * (one counter for the whole chip: 11.7 ms); runs of 8192 and one contiguous range per XCD measure the
     * same, shorter runs lose locality (128: 9.95 ms). */
    const uint32_t run_shift = min(13u, max(6u, 31u - (uint32_t)__clz((int)((n_items >> 6) | 1u))));
    uint32_t it = 0, victim = 0, pend = 0;
    if (lane == 0) pend = atomicAdd(ga.heads + 32u * xcd, 1u);
    for (;;) {
        {
            /* own items first, then those of the XCDs after this one (the tail of the kernel only); the
             * counter is read one item ahead so its round trip to the fabric is off the item's path */
            const uint32_t x = (xcd + victim) & 7u;
            const uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend);
            /* the n-th item of XCD x: runs of 2^run_shift consecutive items go round the XCDs */
            it = ((((nxt >> run_shift) << 3) + x) << run_shift) + (nxt & ((1u << run_shift) - 1u));
            if (it >= n_items) {
                if (++victim == 8u) break;
                if (lane == 0) pend = atomicAdd(ga.heads + 32u * ((xcd + victim) & 7u), 1u);
                continue;
            }
            if (lane == 0) pend = atomicAdd(ga.heads + 32u * x, 1u);
        }
        const uint32_t slot = ga.items[it];
        uint32_t group, chunk;
        split_slot(slot, ga.n_chunks, ga.inv_chunks, group, chunk);
        uint32_t m = (uint32_t)ga.masks[slot];
        /* stage the chunk (1536 floats = six float4 per lane, contiguous) and the group's poses */
        {
            const float4* src = reinterpret_cast<const float4*>(a.spos3 + (size_t)chunk * (kChunkPts * 3));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); /* earlier reads of the LDS arrays are done */
            __builtin_amdgcn_wave_barrier();
            const float4 v0 = src[lane], v1 = src[64 + lane], v2 = src[128 + lane], v3 = src[192 + lane],
                         v4 = src[256 + lane], v5 = src[320 + lane];
            /* float4 f of the pose block = part f % 3 (A, B, C of lds_pose) of pose f / 3; G * 3 of them, 64 per round */
            float4 pv[(G * 3 + 63) / 64];
#pragma unroll
            for (int r = 0; r < (G * 3 + 63) / 64; ++r) {
                const uint32_t f = min(64u * r + lane, (uint32_t)G * 3u - 1u);
                const uint32_t hp = min(group * G + f / 3u, a.n_hyp - 1u), pc = f % 3u;
                /* A, B, C of lds_pose from the column-major 4x4 */
                const float* mp = a.poses + (size_t)hp * 16;
                pv[r] = make_float4(mp[pc == 0u ? 0 : pc == 1u ? 8 : 2], mp[pc == 0u ? 1 : pc == 1u ? 9 : 6],
                                    mp[pc == 0u ? 4 : pc == 1u ? 12 : 10], mp[pc == 0u ? 5 : pc == 1u ? 13 : 14]);
            }
            const float2 ax = ga.aux[min(group * G + (lane % (uint32_t)G), a.n_hyp - 1u)];
            float4* dst = reinterpret_cast<float4*>(pts);
            dst[lane] = v0;
            dst[64 + lane] = v1;
            dst[128 + lane] = v2;
            dst[192 + lane] = v3;
            dst[256 + lane] = v4;
            dst[320 + lane] = v5;
#pragma unroll
            for (int r = 0; r < (G * 3 + 63) / 64; ++r)
                if (64u * r + lane < (uint32_t)G * 3u) poses[64u * r + lane] = pv[r];
            if (lane < (uint32_t)G) auxl[lane] = ax;
            if (BALL && lane < (uint32_t)G) cen[lane] = ca.opos[ca.centers[min(group * G + lane, a.n_hyp - 1u)]];
        }
        const float4 qs = ca.subs[(size_t)chunk * 32u + (lane & 31u)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        /* Which (hypothesis, quarter-tile) pairs are evaluated — two levels. Level 1: the chunk's eight 64-point tiles
         * under the group's sixteen poses, sphere against the voxel box (lane = (tile & 3) * 16 + hypothesis, two
         * rounds): 128 cheap tests. Level 2: only the quarter-tiles of tiles that passed are probed (sphere against the
         * box AND the distance field at its centre), densely: the candidates are listed in quarter-major order and taken
         * 64 at a time. The survivors — the same pairs the one-level scheme of 512 probes per item kept, minus those
         * whose whole tile cannot reach the grid — are compacted in place into the entry list. */
        static_assert(G == 16, "the two-level probe is laid out for groups of 16");
        const uint32_t ql = lane & 31u;
        uint32_t n = 0;
        {
            unsigned long long tb[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t g = lane & 15u;
                const float4 ts = ca.tiles[(size_t)chunk * 8u + 4u * r + (lane >> 4)];
                const pose12 T = lds_pose(poses + 3u * g);
                const float2 ax = auxl[g];
                bool hit = sphere_hits_box(a, T, ax.x, ax.y, ts) & (((m >> g) & 1u) != 0u);
                if (BALL) {
                    const float4 C = cen[g];
                    hit = hit && !(len(make3(ts.x, ts.y, ts.z) - make3(C.x, C.y, C.z)) > ca.ball_r + ts.w * 1.00001f + ball_margin);
                }
                tb[r] = __ballot(hit);
            }
            /* lane q < 32: the hypotheses whose tile q >> 2 passed = a 16-bit field of the round's ballot */
            const uint32_t t = ql >> 2;
            uint32_t cm = (uint32_t)((t & 4u ? tb[1] : tb[0]) >> (16u * (t & 3u))) & 0xFFFFu;
            if (lane >= 32u) cm = 0u;
            const uint32_t mine = (uint32_t)__popc(cm);
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const uint32_t u = __shfl_up(incl, off);
                if (lane >= (uint32_t)off) incl += u;
            }
            const uint32_t nc = __shfl(incl, 31);
            {
                uint32_t o = incl - mine, b = cm;
                while (b) {
                    const uint32_t g = (uint32_t)__builtin_ctz(b);
                    b &= b - 1u;
                    ent[o++] = (unsigned short)((g << 5) | lane);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t i = 0; i < nc; i += 64u) {
                const uint32_t e = i + lane;
                const bool live = e < nc;
                const uint32_t en = ent[live ? e : i];
                const uint32_t g = en >> 5, q = en & 31u;
                const float4 qq = make_float4(__shfl(qs.x, (int)q), __shfl(qs.y, (int)q), __shfl(qs.z, (int)q), __shfl(qs.w, (int)q));
                const pose12 T = lds_pose(poses + 3u * g);
                const float2 ax = auxl[g];
                tile_probe_t pr;
                if (NEAR) {
                    pr = tile_probe(a, T, ax.x, ax.y, ca.tcode, ca.inv_q, qq);
                } else {
                    pr.hit = sphere_hits_box(a, T, ax.x, ax.y, qq);
                    pr.need = false;
                    pr.brick = 0u;
                    pr.lim = 0.f;
                }
                pr.hit = pr.hit & live;
                if (BALL) {
                    const float4 C = cen[g];
                    pr.hit = pr.hit && !(len(make3(qq.x, qq.y, qq.z) - make3(C.x, C.y, C.z)) > ca.ball_r + qq.w * 1.00001f + ball_margin);
                }
                pr.need = pr.need & pr.hit;
                if (!pr.need) pr.brick = 0u;
                const uint32_t code = NEAR ? (uint32_t)a.bcode_tile[pr.brick] : 0u;
                const bool keep = tile_decide(pr, code);
                const unsigned long long sv = __ballot(keep);
                /* in place: the survivors so far are no more than the candidates read so far, and every lane holds its
                 * own candidate in a register before any lane writes */
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (keep) ent[n + (uint32_t)__popcll(sv & ((1ull << lane) - 1ull))] = (unsigned short)en;
                n += (uint32_t)__popcll(sv);
            }
        }
        if (n == 0u || TM_ABLATE_EQ(a, 10)) continue; /* nothing survives (ablate 10 = timing experiment: staging and culling only) */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (STATS) evals += n * 16u / 64u + ((lane < (n * 16u) % 64u) ? 1u : 0u); /* sums to 16 n over the wave */
        auto turn = [&](const uint32_t i, const uint32_t n_end) {
            f3 q[NC];
            uint32_t pidx[NC], gk[NC], lin[NC], mi[NC], brick[NC], code[NC];
            bool near[NC], in[NC];
            /* the half of a turn that ends in gathers runs at a raised priority, the arithmetic that follows them at the
             * normal one: a wave that can put loads in flight goes before one that is adding up terms (-1 %, same-run,
             * gpurun_out/r05_prio*; 1 / 2 / 3 alike, the reverse assignment nothing) */
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const uint32_t e = i + 4u * k + (lane >> 4);
                in[k] = e < n_end;
                const uint32_t en = ent[in[k] ? e : i];
                gk[k] = en >> 5;
                pidx[k] = (en & 31u) * 16u + (lane & 15u);
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const float4* pp = poses + 3u * gk[k];
                const float* sp = pts + 3u * pidx[k];
                if (BALL) { /* exact membership of the neighbourhood, on the untransformed point */
                    const float4 C = cen[gk[k]];
                    in[k] = in[k] & (len2(make3(sp[0], sp[1], sp[2]) - make3(C.x, C.y, C.z)) <= ca.ball_r2);
                }
                q[k] = lds_xform_point(pp[0], pp[1], pp[2], sp[0], sp[1], sp[2]);
                const f2v vxy = f2v{a.g.sx, a.g.sy} * f2v{q[k].x, q[k].y} + f2v{a.g.tx, a.g.ty};
                const float vx = vxy.x, vy = vxy.y, vz = a.g.sz * q[k].z + a.g.tz;
                in[k] = in[k] & (vx > -1.f) & (vx < a.g.fex) & (vy > -1.f) & (vy < a.g.fey) & (vz > -1.f) &
                        (vz < a.g.fez);
                uint32_t ix = in[k] ? (uint32_t)(int)vx : 0u, iy = in[k] ? (uint32_t)(int)vy : 0u,
                         iz = in[k] ? (uint32_t)(int)vz : 0u;
                brick[k] = brick_of(ix, iy, iz, (uint32_t)a.bx, (uint32_t)a.by);
                lin[k] = brick[k] * 64u + (((iz & 3u) << 4) | ((iy & 3u) << 2) | (ix & 3u));
                code[k] = NEAR ? (uint32_t)a.bcode[TM_ABLATE_BIT(a, 8) ? (brick[k] & 1023u) : brick[k]] : 0u;
                if (TM_ABLATE_BIT(a, 8)) code[k] = 0u; /* the folded code is another brick's: every in-grid point is near */
                __builtin_amdgcn_sched_barrier(0); /* one pose in registers at a time */
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                near[k] = in[k] & (code[k] <= a.code_max);
                if (STATS) {
                    n_in += in[k] ? 1u : 0u;
                    n_near += near[k] ? 1u : 0u;
                }
                lin[k] = near[k] ? lin[k] : 0u;
                if (TM_ABLATE_EQ(a, 7)) lin[k] &= 0x3FFFFu; /* timing experiment: voxel gathers stay in 1 MB */
                /* timing experiments (wrong results): the access pattern of a table with 2-byte / 1-byte entries — a
                 * brick is one 128-B line / half a line, the touched shell a half / a quarter of today's */
                if (TM_ABLATE_EQ(a, 12)) lin[k] = near[k] ? ((brick[k] << 5) | (lin[k] & 31u)) : 0u;
                if (TM_ABLATE_EQ(a, 13)) lin[k] = near[k] ? ((brick[k] << 4) | (lin[k] & 15u)) : 0u;
                if (TM_ABLATE_BIT(a, 2)) lin[k] &= 0x3FFFu; /* 64 KB window */
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) mi[k] = W32 ? ld_off32(a.vox, lin[k] << 2) : a.vox[lin[k]];
            if (TM_ABLATE_BIT(a, 4)) {
#pragma unroll
                for (int k = 0; k < NC; ++k) mi[k] &= 1023u;
            }
            /* the model position and, for the near points, both orientation records (scene point's, model point's) in
             * ONE level: mi is known, the scene point always was. The orientation term then starts without a round
             * trip of its own (kernel 7.72 -> 7.39 ms on the bench batch); the fifth of the near points that fail the
             * distance test fetch their two records for nothing. */
            float mpx[NC][3];
            float4 sr[NC], mr[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (W32) {
                    const f3pack t = ld_off32(reinterpret_cast<const f3pack*>(a.mpos3), mi[k] * 12u);
                    mpx[k][0] = t.x, mpx[k][1] = t.y, mpx[k][2] = t.z;
                } else {
                    const float* src = a.mpos3 + 3u * (size_t)mi[k];
                    mpx[k][0] = src[0], mpx[k][1] = src[1], mpx[k][2] = src[2];
                }
            }
            if (!TM_ABLATE_BIT(a, 1)) {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    if (near[k]) {
                        if (W32) {
                            sr[k] = ld_off32(a.sref, ((TM_ABLATE_BIT(a, 32) ? 0u : chunk * (uint32_t)kChunkPts) + pidx[k]) << 4);
                            mr[k] = ld_off32(a.mref, (TM_ABLATE_BIT(a, 16) ? (mi[k] & 1023u) : mi[k]) << 4);
                        } else {
                            sr[k] = a.sref[chunk * (uint32_t)kChunkPts + pidx[k]];
                            mr[k] = a.mref[mi[k]];
                        }
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            bool ok[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (L2) {
                    float dx = q[k].x - mpx[k][0], dy = q[k].y - mpx[k][1], dz = q[k].z - mpx[k][2];
                    float d2 = (dx * dx + dy * dy) + dz * dz;
                    ok[k] = near[k] & (d2 <= a.d2max);
                } else {
                    /* !(sqrtf(d2) > thres) without the square root: sqrtf is correctly rounded and monotone and
                     * d2max is the largest float whose root is <= thres, so sqrtf(d2) > thres <=> d2 > d2max; a
                     * NaN fails both comparisons alike. The exhaustive kernel keeps the literal form (eval_point). */
                    float d2 = traits_distance2(a.kind, a.radius, q[k], make3(mpx[k][0], mpx[k][1], mpx[k][2]));
                    ok[k] = near[k] & !(d2 > a.d2max);
                }
                if (TM_ABLATE_BIT(a, 6)) ok[k] = near[k] & ((pidx[k] & 7u) != 0u); /* folded entries fail the test: let 7 of 8 through */
            }
            if (TM_ABLATE_EQ(a, 1) || TM_ABLATE_EQ(a, 7) || TM_ABLATE_EQ(a, 12) || TM_ABLATE_EQ(a, 13) || TM_ABLATE_BIT(a, 1)) {
#pragma unroll
                for (int k = 0; k < NC; ++k) pass += ok[k] ? 1u : 0u; /* timing experiments: no orientation stage */
                return;
            }
            /* orientation term (:441-488): every correspondence adds count << 52 | 2^-40 fixed-point term to its
             * hypothesis' sum in LDS, one atomic per lane (the 16-lane DPP reductions of two dwords that used to stand
             * here were a third of the loop's VALU instructions for the same integers) */
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (ok[k]) {
                    if (STATS) ++pass;
                    if (sr[k].w == mr[k].w) {
                        const float4* pp = poses + 3u * gk[k];
                        f3 r = lds_xform_dir(pp[0], pp[1], pp[2], sr[k].x, sr[k].y, sr[k].z);
                        float term = fabsf(dot3(r, make3(mr[k].x, mr[k].y, mr[k].z)));
                        atomicAdd(&acc[gk[k]], (1ull << 52) + fx40_trunc(term));
                    }
                }
            }
        };
        for (uint32_t i = 0; i < n; i += 4u * NC) turn(i, n);
        /* flush the item's sums: one pair of atomics per hypothesis with correspondences */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < (uint32_t)G) {
            const unsigned long long v = acc[lane];
            if (v && !TM_ABLATE_EQ(a, 11)) {
                const uint32_t h = group * G + lane;
                atomicAdd(&a.counts[h], (uint32_t)(v >> 52));
                atomicAdd(&a.scores_fx[h], v & 0xFFFFFFFFFFFFFull);
                acc[lane] = 0ull;
            }
        }
    }
    if (!STATS) return;
    for (int off = 32; off > 0; off >>= 1) {
        evals += __shfl_down(evals, off);
        pass += __shfl_down(pass, off);
        n_in += __shfl_down(n_in, off);
        n_near += __shfl_down(n_near, off);
    }
    __syncthreads();
    uint32_t(*s_st)[4] = reinterpret_cast<uint32_t(*)[4]>(s_raw[0]); /* the loop is over: reuse wave 0's area */
    if (lane == 0) {
        s_st[wave][0] = evals;
        s_st[wave][1] = pass;
        s_st[wave][2] = n_in;
        s_st[wave][3] = n_near;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        a.stat_slots[blockIdx.x] = make_uint4((s_st[0][0] + s_st[1][0]) + (s_st[2][0] + s_st[3][0]),
                                              (s_st[0][1] + s_st[1][1]) + (s_st[2][1] + s_st[3][1]),
                                              (s_st[0][2] + s_st[1][2]) + (s_st[2][2] + s_st[3][2]),
                                              (s_st[0][3] + s_st[1][3]) + (s_st[2][3] + s_st[3][3]));
}

/* the instantiation for a launch: metric filter / neighbourhood restriction / group size / 32-bit gathers */
using items_kernel_t = void (*)(score_args, cull_args, group_args);
template <bool STATS>
static items_kernel_t items_kernel(bool l2, bool ball, bool near_metric, bool w32) {
    if (near_metric)
        return ball ? k_score_items<false, 4, 16, true, true, false, STATS>
               : w32 ? k_score_items<false, 4, 16, false, true, true, STATS>
                     : k_score_items<false, 4, 16, false, true, false, STATS>;
    if (ball) return l2 ? k_score_items<true, 4, 16, true, true, false, STATS> : k_score_items<false, 4, 16, true, false, false, STATS>;
    if (w32) return l2 ? k_score_items<true, 4, 16, false, true, true, STATS> : k_score_items<false, 4, 16, false, false, true, STATS>;
    return l2 ? k_score_items<true, 4, 16, false, true, false, STATS> : k_score_items<false, 4, 16, false, false, false, STATS>;
}

/* <<< scoring kernels */

/* sums the per-workgroup stat slots of one scoring launch into that launch's row of device counters
 * (one workgroup, plain stores: nothing for the host to wait for inside a timed loop) */
__global__ __launch_bounds__(1024) void k_stats_reduce(const uint4* __restrict__ slots, uint32_t n_slots,
                                                       const uint32_t* __restrict__ n_items,
                                                       unsigned long long* __restrict__ row) {
    __shared__ unsigned long long s_acc[16][4];
    unsigned long long acc[4] = {0, 0, 0, 0};
    for (uint32_t i = threadIdx.x; i < n_slots; i += 1024) {
        const uint4 v = slots[i];
        acc[0] += v.x; /* evaluations */
        acc[1] += v.y; /* distance-passing */
        acc[2] += v.z; /* in-grid */
        acc[3] += v.w; /* near */
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s_acc[threadIdx.x >> 6][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned long long r = 0;
        for (int w = 0; w < 16; ++w) r += s_acc[w][threadIdx.x];
        row[threadIdx.x] = r;
    }
    if (threadIdx.x == 4) row[4] = n_items ? *n_items : 0u; /* grouped kernel: chunks staged into LDS */
}

/* subset flags (original order) -> sorted order of the scoring arrays */
__global__ __launch_bounds__(256) void k_active_sorted(const unsigned char* __restrict__ active,
                                                       const uint32_t* __restrict__ order, uint32_t n, uint32_t n_pad,
                                                       unsigned char* __restrict__ out) {
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n_pad) out[k] = k < n ? active[order[k]] : 0;
}

__global__ void k_finalize_scores(const unsigned long long* __restrict__ fx, uint32_t n, double inv_scale,
                                  double model_size, double* __restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = ((double)fx[t] * inv_scale) / model_size; /* reference :509 */
}

/* finish_find for several poses side by side (blockIdx.y = pose k): per original index the matched model index or
 * 0xFFFFFFFF, the matches of pose k in match_of_orig + k * n,
 * count and fixed-point score sum in count_fx[2 k] / [2 k + 1] */
__global__ __launch_bounds__(256) void k_extract_batch(score_args a, uint32_t n_pad, uint32_t n,
                                                       uint32_t* __restrict__ match_of_orig,
                                                       unsigned long long* __restrict__ count_fx,
                                                       const float4* __restrict__ tiles = nullptr, float coord_max = 0.f) {
    __shared__ unsigned long long s_fx[4];
    __shared__ uint32_t s_cnt[4];
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    const uint32_t k = blockIdx.y;
    const pose12 T = load_pose(a.poses + 16 * (size_t)k);
    /* a wave = a 64-point tile: one that cannot reach the model's voxel box under this pose holds no match */
    bool reach = s < n_pad;
    if (reach && tiles) {
        const float2 ax = pose_aux(T, coord_max);
        reach = sphere_hits_box(a, T, ax.x, ax.y, tiles[s >> 6]);
    }
    if (!__syncthreads_or(reach ? 1 : 0)) return; /* none of the workgroup's four tiles: nothing to add either */
    const float4 v = reach ? a.spos[s] : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
    long long fx = 0;
    uint32_t hit = 0;
    if (v.x == v.x) {
        uint32_t pass = 0, m = 0xFFFFFFFFu;
        hit = eval_point(a, T, make3(v.x, v.y, v.z), s, fx, pass, &m);
        if (hit) match_of_orig[(size_t)k * n + __float_as_uint(v.w)] = m;
    }
    /* count and score sum (integers: any order) per workgroup, then one pair of atomics — one pair per match put
     * ~2e5 atomics per pose on two addresses, 0.17 ms of a 4-pose call */
    unsigned long long w = hit ? (unsigned long long)fx : 0ull;
    for (int off = 32; off > 0; off >>= 1) w += (unsigned long long)__shfl_down((long long)w, off);
    const uint32_t c = (uint32_t)__popcll(__ballot(hit != 0u));
    if ((threadIdx.x & 63u) == 0u) {
        s_fx[threadIdx.x >> 6] = w;
        s_cnt[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t cb = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
        if (cb) {
            atomicAdd(&count_fx[2 * (size_t)k + 1], (s_fx[0] + s_fx[1]) + (s_fx[2] + s_fx[3]));
            atomicAdd(&count_fx[2 * (size_t)k], (unsigned long long)cb);
        }
    }
}
/* ordered compaction of those matches: ascending scene index = the reference's scan order (finish_find walks 0..n-1,
 * include/impl/scene.hpp:100-106). Per pose: matches per 64 original indices (row k of wave_count, n_waves + 1 entries each,
 * the last one zero) */
__global__ __launch_bounds__(256) void k_extract_count_batch(const uint32_t* __restrict__ match_of_orig, uint32_t n,
                                                             uint32_t n_waves, uint32_t* __restrict__ wave_count) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    const bool hit = i < n && match_of_orig[(size_t)k * n + i] != 0xFFFFFFFFu;
    const unsigned long long b = __ballot(hit);
    if ((threadIdx.x & 63) == 0) wave_count[(size_t)k * (n_waves + 1u) + (i >> 6)] = (uint32_t)__popcll(b);
    if (blockIdx.x == 0 && threadIdx.x == 0) wave_count[(size_t)k * (n_waves + 1u) + n_waves] = 0u; /* the row's spare entry */
}
/* ... and the ordered lists: wave_offset = exclusive scan over ALL rows of wave_count, so a pose's entries start at
 * its row's first offset */
__global__ __launch_bounds__(256) void k_extract_scatter_batch(const uint32_t* __restrict__ match_of_orig, uint32_t n,
                                                               uint32_t n_waves, const uint32_t* __restrict__ wave_offset,
                                                               const uint32_t* __restrict__ new2orig,
                                                               uint32_t* __restrict__ corr /* rows of 2 n: scene, model */) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    const uint32_t mt = i < n ? match_of_orig[(size_t)k * n + i] : 0xFFFFFFFFu;
    const bool hit = mt != 0xFFFFFFFFu;
    const unsigned long long b = __ballot(hit);
    if (!hit) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t rank = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    const uint32_t* row = wave_offset + (size_t)k * (n_waves + 1u);
    const uint32_t dst = row[i >> 6] - row[0] + rank;
    corr[2 * (size_t)n * k + dst] = i;
    corr[2 * (size_t)n * k + n + dst] = new2orig[mt];
}

/* finish_find + cross-covariance sums for ICP. Grid-stride over the sorted scene points; the 15
 * fixed-point sums and the count are integers, so thread-local accumulation, the wave shuffles, the LDS
 * step and the final pass may add them in any order. Each block stores its 16 partials; k_icp_finalize
 * adds the blocks (no atomics: sixteen same-line atomics per wave serialise in one L2 channel). */
constexpr int kIcpBlocks = 128; /* per candidate: 512 waves deal the scene's tiles among themselves; the epilogue (16 64-bit sums
                                 * through the wave, 400 cross-lane moves) is paid per wave, and was most of the kernel with 2048 */

struct icp_state;
__device__ __forceinline__ const uint32_t* icp_done_word(const icp_state* st, uint32_t cand);

/* blockIdx.y = candidate (tm_icp_refine_batch: the candidates of one find_parallel round are refined side by side):
 * pose a.poses + 16 y, partials [y][gridDim.x][16], loop state st[y] (null on the first pass) */
__global__ __launch_bounds__(256) void k_icp_accumulate(score_args a, uint32_t n_pad,
                                                        unsigned long long* __restrict__ partials /* [cand][grid][16] */,
                                                        const icp_state* __restrict__ st = nullptr,
                                                        const float4* __restrict__ tiles = nullptr, float coord_max = 0.f) {
    __shared__ long long s_acc[4][16];
    const uint32_t cand = blockIdx.y;
    if (st && *icp_done_word(st, cand)) return; /* tm_icp_refine: this candidate's loop has ended */
    a.poses += (size_t)cand * 16;
    partials += (size_t)cand * gridDim.x * 16;
    long long acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0;
    const pose12 T = load_pose(a.poses);
    const float2 ax = pose_aux(T, coord_max);
    /* a wave = a 64-point tile of the sorted scene at a time, tiles dealt round the waves of the grid: a tile whose
     * bounding sphere cannot reach the model's voxel box under this pose (the conservative test of the scoring kernels)
     * holds no in-grid point and is passed over — a candidate touches a tenth of the scene, and its tiles, neighbours in
     * the curve order, spread evenly over the waves */
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < n_pad; s += gridDim.x * 256) {
        if (tiles && !sphere_hits_box(a, T, ax.x, ax.y, tiles[s >> 6])) continue;
        float4 v = a.spos[s];
        if (!(v.x == v.x)) continue;
        long long fx = 0;
        uint32_t pass = 0, m = 0;
        if (!eval_point(a, T, make3(v.x, v.y, v.z), s, fx, pass, &m)) continue;
        float4 mp = a.mpos[m];
        const double sc[3] = {(double)v.x, (double)v.y, (double)v.z};
        const double mc[3] = {(double)mp.x, (double)mp.y, (double)mp.z};
        const double S = 4294967296.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc[c] += (long long)(sc[c] * S);
            acc[3 + c] += (long long)(mc[c] * S);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[6 + 3 * r + c] += (long long)((sc[r] * mc[c]) * S);
        acc[15] += 1;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) s_acc[threadIdx.x >> 6][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 16)
        partials[(size_t)blockIdx.x * 16 + threadIdx.x] = (unsigned long long)(
            s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x] + s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]);
}

/* One turn of icp_ (reference include/impl/scene.hpp:376-400) on the device, a single thread: `sums` holds the
 * correlation sums of the pose in `pose` (the initial one when first, else the last fit). State: best sums /
 * count / pose, iteration count, done flag. When the loop goes on, the next fit is written to `pose`. */
struct icp_state {
    unsigned long long best_sums[16]; /* [15] = count */
    float best_t[16];
    uint32_t iter, done;
};
__device__ __forceinline__ const uint32_t* icp_done_word(const icp_state* st, uint32_t cand) { return &st[cand].done; }
__device__ void icp_step(const unsigned long long* sums, float* pose, icp_state* st, uint32_t max_iterations, bool first) {
    if (first) {
        st->iter = 0;
        st->done = 0;
    } else {
        if (st->done) return;
        if ((uint32_t)sums[15] < (uint32_t)st->best_sums[15]) { /* :396: fewer correspondences than before */
            st->done = 1;
            return;
        }
    }
    for (int k = 0; k < 16; ++k) st->best_sums[k] = sums[k];
    for (int k = 0; k < 16; ++k) st->best_t[k] = pose[k];
    if (!first && ++st->iter == max_iterations) {
        st->done = 1;
        return;
    }
    if ((uint32_t)sums[15] < 3u || max_iterations == 0u) { /* :381 */
        st->done = 1;
        return;
    }
    long long v[15];
    for (int k = 0; k < 15; ++k) v[k] = (long long)sums[k];
    float T[16];
    rigid_fit_sums(v, (uint32_t)sums[15], T); /* :385-393 */
    for (int k = 0; k < 16; ++k) pose[k] = T[k];
}

/* st != null (tm_icp_refine): thread 0 goes on with the turn of the loop these sums belong to. One workgroup per
 * candidate (blockIdx.x): partials [cand][n_blocks][16], sums [cand][16], pose [cand][16], st[cand]. */
constexpr int kIcpFinalizeThreads = 1024; /* 64 slices of the block list x 16 sums: 8 loads per thread for 512 blocks */
__global__ __launch_bounds__(kIcpFinalizeThreads) void k_icp_finalize(const unsigned long long* __restrict__ partials, uint32_t n_blocks,
                                                      unsigned long long* __restrict__ sums /* [cand][16] */,
                                                      int check_done = 0,
                                                      float* pose = nullptr, icp_state* st = nullptr,
                                                      uint32_t max_iterations = 0, int first = 0) {
    constexpr uint32_t kSlices = kIcpFinalizeThreads / 16;
    __shared__ unsigned long long s_part[kSlices][16];
    const uint32_t cand = blockIdx.x;
    partials += (size_t)cand * n_blocks * 16;
    sums += (size_t)cand * 16;
    if (pose) pose += (size_t)cand * 16;
    if (st) st += cand;
    if (check_done && st && st->done) return;
    const uint32_t k = threadIdx.x & 15, slice = threadIdx.x >> 4; /* slices of the block list */
    unsigned long long t = 0;
    for (uint32_t b = slice; b < n_blocks; b += kSlices) t += partials[(size_t)b * 16 + k];
    s_part[slice][k] = t;
    __syncthreads();
    if (threadIdx.x < 64) { /* four lanes per sum, 16 slices each, then two shuffles */
        const uint32_t kk = threadIdx.x & 15, q = threadIdx.x >> 4;
        unsigned long long r = 0;
        for (uint32_t sl = q * (kSlices / 4); sl < (q + 1) * (kSlices / 4); ++sl) r += s_part[sl][kk];
        r += (unsigned long long)__shfl_down((long long)r, 32);
        r += (unsigned long long)__shfl_down((long long)r, 16);
        if (threadIdx.x < 16) sums[threadIdx.x] = r;
    }
    if (st) {
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x == 0) icp_step(sums, pose, st, max_iterations, first != 0);
    }
}

/* ------------------------------------------------------------------------------------------------
 * hypothesis generation
 * ---------------------------------------------------------------------------------------------- */
struct gen_args {
    const float4* opos;
    const float4* otan;
    uint32_t n_scene;
    const float4* mpos;
    const float4* mtan;
    const uint32_t* ht_slot;
    const uint4* ht_key;
    uint32_t cap_mask;
    const uint32_t* seg_begin;
    const uint2* entries;       /* NEW model indices */
    const uint32_t* new2orig;   /* model: internal -> original index */
    float lower2, upper2; /* squared diameter window, reference include/impl/scene.hpp:117-120 */
    float lo0, hi0;
    uint32_t steps;
    float angle_step;
    uint32_t limit;
    int force_up;
};

/* one thread per scene sample pair: filters (:290-297), feature/valid (:299-302), query (:304) */
__global__ void k_gen_lookup(gen_args g, const uint2* __restrict__ ij, uint32_t n_pairs,
                             uint32_t* __restrict__ seg_first, uint32_t* __restrict__ take) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pairs) return;
    uint2 p = ij[t];
    uint32_t tk = 0, first = 0;
    if (p.x < g.n_scene && p.y < g.n_scene && p.x != p.y) {
        float4 P1 = g.opos[p.x], T1 = g.otan[p.x];
        float4 P2 = g.opos[p.y], T2 = g.otan[p.y];
        bool ok = (T1.w == 0.f) && (P2.w != 0.f) && (T2.w == 0.f); /* !mask[i], tangent_mask[j], !mask[j] */
        if (ok) {
            f3 p1 = make3(P1.x, P1.y, P1.z), p2 = make3(P2.x, P2.y, P2.z);
            f3 t1 = make3(T1.x, T1.y, T1.z), t2 = make3(T2.x, T2.y, T2.z);
            f3 d = p2 - p1;
            float sq = len2(d);
            d = unit(d);
            ok = !(sq < g.lower2 || sq > g.upper2);
            ok = ok && !(1.f - fabsf(dot3(d, t1)) < 0.01f);
            if (ok) {
                feat4 f = pair_feature(p1, t1, p2, t2);
                if (feature_valid(f, g.lo0, g.hi0)) {
                    key4 k = feature_key(f, g.lo0, g.hi0, g.steps, g.angle_step);
                    uint32_t s = ht_find(g.ht_slot, g.ht_key, g.cap_mask, k);
                    if (s != 0xFFFFFFFFu) {
                        first = g.seg_begin[s];
                        tk = min(g.seg_begin[s + 1] - first, g.limit);
                    }
                }
            }
        }
    }
    seg_first[t] = first;
    take[t] = tk;
}

/* one thread per hypothesis slot, slots in visit order (pair-major, table order inside a pair) */
__global__ void k_gen_emit(gen_args g, const uint2* __restrict__ ij, uint32_t n_pairs,
                           const uint32_t* __restrict__ seg_first, const uint32_t* __restrict__ offs,
                           uint32_t total, uint32_t cap, float* __restrict__ poses,
                           uint4* __restrict__ prov, uint32_t* __restrict__ keep) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    /* last pair with offs[pair] <= t */
    uint32_t lo = 0, hi = n_pairs;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (offs[mid] <= t) lo = mid; else hi = mid;
    }
    uint32_t e = t - offs[lo];
    uint2 p = ij[lo];
    uint2 mm = g.entries[seg_first[lo] + e];
    float4 P1 = g.opos[p.x], T1 = g.otan[p.x], P2 = g.opos[p.y];
    float4 Mi = g.mpos[mm.x], Mj = g.mpos[mm.y], Ti = g.mtan[mm.x];
    float T[16];
    base_transform(make3(P1.x, P1.y, P1.z), make3(P2.x, P2.y, P2.z), make3(T1.x, T1.y, T1.z),
                   make3(Mi.x, Mi.y, Mi.z), make3(Mj.x, Mj.y, Mj.z), make3(Ti.x, Ti.y, Ti.z), T);
    uint32_t ok = (g.force_up && fabsf(1.f - T[10]) > 0.01f) ? 0u : 1u; /* reference :317 */
    if (keep) keep[t] = ok;
    if (t < cap || keep) {
        float4* dst = (float4*)(poses + (size_t)t * 16);
        dst[0] = make_float4(T[0], T[1], T[2], T[3]);
        dst[1] = make_float4(T[4], T[5], T[6], T[7]);
        dst[2] = make_float4(T[8], T[9], T[10], T[11]);
        dst[3] = make_float4(T[12], T[13], T[14], T[15]);
        prov[t] = make_uint4(lo, g.new2orig[mm.x], g.new2orig[mm.y], e);
    }
}

__global__ void k_compact(const float4* __restrict__ poses_in, const uint4* __restrict__ prov_in,
                          const uint32_t* __restrict__ keep, const uint32_t* __restrict__ dst_of,
                          uint32_t total, uint32_t cap, float4* __restrict__ poses_out,
                          uint4* __restrict__ prov_out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total || !keep[t]) return;
    uint32_t d = dst_of[t];
    if (d >= cap) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) poses_out[(size_t)d * 4 + k] = poses_in[(size_t)t * 4 + k];
    prov_out[d] = prov_in[t];
}

/* argmax over a scored batch: count desc, score desc, id asc. One workgroup. */
__global__ __launch_bounds__(1024) void k_best(const uint32_t* __restrict__ counts,
                                               const double* __restrict__ scores,
                                               const float* __restrict__ poses, uint32_t n,
                                               uint32_t id_offset, uint32_t id_stride,
                                               tm_best_record* __restrict__ out) {
    __shared__ uint32_t s_cnt[1024];
    __shared__ double s_sc[1024];
    __shared__ uint32_t s_idx[1024];
    uint32_t bc = 0, bi = 0xFFFFFFFFu;
    double bs = -1.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        uint32_t c = counts[i];
        double s = scores[i];
        bool better = bi == 0xFFFFFFFFu || c > bc || (c == bc && (s > bs || (s == bs && i < bi)));
        if (better) {
            bc = c;
            bs = s;
            bi = i;
        }
    }
    s_cnt[threadIdx.x] = bc;
    s_sc[threadIdx.x] = bs;
    s_idx[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            uint32_t c = s_cnt[threadIdx.x + off], i = s_idx[threadIdx.x + off];
            double s = s_sc[threadIdx.x + off];
            uint32_t mc = s_cnt[threadIdx.x], mi = s_idx[threadIdx.x];
            double ms = s_sc[threadIdx.x];
            bool better = i != 0xFFFFFFFFu &&
                          (mi == 0xFFFFFFFFu || c > mc || (c == mc && (s > ms || (s == ms && i < mi))));
            if (better) {
                s_cnt[threadIdx.x] = c;
                s_sc[threadIdx.x] = s;
                s_idx[threadIdx.x] = i;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tm_best_record r;
        uint32_t i = s_idx[0];
        r.inlier_count = i == 0xFFFFFFFFu ? 0u : s_cnt[0];
        r.hyp_id = i == 0xFFFFFFFFu ? 0xFFFFFFFFu : id_offset + i * id_stride;
        r.score = i == 0xFFFFFFFFu ? 0.0 : s_sc[0];
        for (int k = 0; k < 16; ++k) r.pose[k] = i == 0xFFFFFFFFu ? 0.f : poses[(size_t)i * 16 + k];
        for (int k = 0; k < 4; ++k) r.pad[k] = 0;
        *out = r;
    }
}

/* ---- host helpers ---- */
static inline bool fin3(const float* p) {
    return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
}

static int check_pair(const tm_scene* s, const tm_model* m) {
    TM_REQUIRE(s && m, TM_ERR_INVALID, "null scene or model");
    TM_REQUIRE(s->ctx == m->ctx, TM_ERR_INVALID, "scene and model belong to different contexts");
    return TM_OK;
}

/* launches the scoring kernels for poses already on the device; results stay on the device */
static int score_device(tm_scene* s, tm_model* m, int traits_kind, const float* d_poses, uint32_t n_hyp,
                        const unsigned char* d_active, uint64_t n_active, float dist_thres, int mode,
                        uint32_t* d_counts, double* d_scores, const uint32_t* d_centers = nullptr,
                        float ball_radius = 0.f) {
    TM_FORWARD(check_pair(s, m));
    TM_REQUIRE(traits_kind == m->metric_kind, TM_ERR_INVALID,
               "traits_kind differs from the model's metric (tm_model_set_metric)");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    TM_REQUIRE(mode == TM_SCORE_AUTO || mode == TM_SCORE_BRUTEFORCE || mode == TM_SCORE_CULLED,
               TM_ERR_INVALID, "bad scoring mode");
    if (n_hyp == 0) return TM_OK;
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    /* The grouped kernels keep two words per (group of hypotheses, 512-point chunk) slot. A large batch on a
     * large scene (1M hypotheses x 10M points = 1.3e9 slots = 10 GB) is therefore scored in several launches
     * of whole groups, each within the slot budget (score_slot_mib, 256 MiB); the sums are integers per
     * hypothesis, so the split changes nothing in the results. */
    const size_t slot_budget = (size_t)c->opt.score_slot_mib * (1u << 20) / (2 * sizeof(uint32_t));
    if (mode != TM_SCORE_BRUTEFORCE && !d_active && c->opt.score_grouped && !(d_centers && c->opt.ball_ungrouped)) {
        const uint32_t gh = 16u;
        const size_t n_chunks = s->n_sorted / kChunkPts;
        const size_t groups = (n_hyp + gh - 1) / gh;
        const size_t groups_per = n_chunks ? slot_budget / n_chunks : groups;
        if (groups * n_chunks > slot_budget && groups_per >= 1) {
            const size_t step = std::min<size_t>(groups_per * gh, 0x7FFFFFF0u);
            for (size_t off = 0; off < n_hyp; off += step)
                TM_FORWARD(score_device(s, m, traits_kind, d_poses + off * 16, (uint32_t)std::min<size_t>(step, n_hyp - off),
                                        d_active, n_active, dist_thres, mode, d_counts + off, d_scores + off,
                                        d_centers ? d_centers + off : nullptr, ball_radius));
            return TM_OK;
        }
    }
    TM_FORWARD(s->w_scores_fx.ensure(n_hyp));
    TM_HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)n_hyp * 4, c->stream));
    TM_HIP_CHECK(hipMemsetAsync(s->w_scores_fx.p, 0, (size_t)n_hyp * 8, c->stream));
    score_args a;
    a.spos = s->spos.p;
    a.spos3 = s->spos3.p;
    a.sref = s->sref.p;
    a.mpos = m->pos.p;
    a.mpos3 = m->pos3.p;
    a.mref = m->ref.p;
    a.vox = m->voxels.p;
    a.active = d_active;
    a.poses = d_poses;
    a.counts = d_counts;
    a.scores_fx = s->w_scores_fx.p;
    a.g = m->grid;
    const float thres = dist_thres * m->resolution; /* reference :413 */
    a.d2max = sq_threshold(thres);
    a.n_hyp = n_hyp;
    a.ablate = c->opt.ablate;
    a.kind = m->metric_kind;
    a.radius = m->metric_radius;
    a.dmax = thres;
    /* brick filter: a voxel hit can be skipped when every voxel centre of its 4x4x4 brick is farther
     * than thres + (largest distance from a voxel's centre to a point of its cell = sqrt(3) cells,
     * the int cast truncates) from the model: then no model point is within thres of the query. */
    const float cell_max = std::max(std::max(1.f / m->grid.sx, 1.f / m->grid.sy), 1.f / m->grid.sz);
    float slack = 1.7320508f * cell_max * 1.0001f + thres * 1e-5f + 1e-7f;
    /* the intrinsic distances are formed from coordinate differences (and 2 pi r): their f32 rounding scales with
     * the coordinates' magnitude, not with the distance */
    if (m->metric_kind > TM_TRAITS_PLANE)
        slack += 1e-5f * (m->metric_coord_max + 6.2831855f * m->metric_radius + 8.f * cell_max);
    /* the distance-field tile test and the brick filter reason with 3-D Euclidean distances: L2 metrics
     * only; the other metrics keep the sphere-vs-box culling */
    const bool culled = mode != TM_SCORE_BRUTEFORCE;
    const bool l2 = m->metric_kind <= TM_TRAITS_PLANE;
    /* plane2 / cylinder: the same two filters with the codes built under the model's metric */
    a.bx = m->bdim[0];
    a.by = m->bdim[1];
    {
        float cm = std::floor((thres + slack) / m->brick_q) + 1.f;
        a.code_max = (thres >= 0.f && cm < 255.f) ? (uint32_t)cm : 255u;
    }
    bool near_metric = !l2 && c->opt.near_metric && m->brick_code_metric_pt.p && m->brick_code_metric_tile.p;
    if (near_metric && c->opt.near_metric == 1) { /* auto: worth it only where there is something to remove */
        uint64_t all = 0, far = 0;
        for (uint32_t k = 0; k < 256u; ++k) {
            all += m->metric_code_hist[k];
            if (k > a.code_max) far += m->metric_code_hist[k];
        }
        near_metric = all > 0 && far * 10 >= all;
    }
    a.bcode = !culled ? nullptr : l2 ? m->brick_code.p : near_metric ? m->brick_code_metric_pt.p : nullptr;
    a.bcode_tile = !culled ? nullptr : l2 ? m->brick_code.p : near_metric ? m->brick_code_metric_tile.p : nullptr;
    const uint32_t tiles = s->n_sorted / kScoreTile;
    /* aim for >= ~8 workgroups per CU; keep chunks long enough to amortise the tile load */
    uint32_t want_chunks = std::max(1u, (256u * 16u + tiles - 1) / tiles);
    uint32_t hpb = std::max(16u, std::min(1024u, (n_hyp + want_chunks - 1) / want_chunks));
    a.hyp_per_block = hpb;
    dim3 grid(tiles, (n_hyp + hpb - 1) / hpb);
    TM_REQUIRE(grid.y <= 65535u, TM_ERR_UNSUPPORTED, "hypothesis batch too large for one launch");
    const size_t n_slots = culled ? (size_t)n_hyp : (size_t)grid.x * grid.y;
    size_t n_stat_slots = n_slots; /* slots the launched kernel actually writes */
    bool stats_kept = true;        /* false: the launched kernel is an instantiation without stage counters */
    const uint32_t* d_n_items = nullptr; /* grouped kernel: device count of staged (group, chunk) items */
    TM_FORWARD(s->w_stats.ensure(n_slots));
    a.stat_slots = s->w_stats.p;
    const uint32_t trow = (uint32_t)(c->n_timed % kTimedRing);
    if (c->timing) TM_HIP_CHECK(hipEventRecord(c->ev_ring[2 * trow], c->stream));
    if (culled) {
        cull_args ca;
        ca.tiles = s->tiles.p;
        ca.supers = s->supers.p;
        ca.subs = s->subs.p;
        ca.n_tiles = s->n_tiles;
        ca.coord_max = s->coord_max;
        ca.inv_q = 1.f / m->brick_q;
        ca.centers = d_centers;
        ca.opos = s->opos.p;
        ca.ball_r = ball_radius;
        ca.ball_r2 = ball_radius * ball_radius;
        ca.tcode = (thres + slack) / m->brick_q + 1.f;
        if (!(thres >= 0.f)) ca.tcode = 1e30f;
        ca.active_sorted = nullptr;
        if (d_active) {
            TM_FORWARD(s->w_active_sorted.ensure(s->n_sorted));
            hipLaunchKernelGGL(k_active_sorted, dim3(s->n_sorted / 256), dim3(256), 0, c->stream, d_active,
                               s->d_order.p, s->n, s->n_sorted, s->w_active_sorted.p);
            ca.active_sorted = s->w_active_sorted.p;
        }
        bool grouped = !d_active; /* with or without the neighbourhood restriction (d_centers) */
        grouped = grouped && c->opt.score_grouped != 0;
        if (d_centers && c->opt.ball_ungrouped) grouped = false;
        /* groups of consecutive hypotheses share the scene chunks they reach */
        group_args ga;
        ga.chunks = s->chunks.p;
        ga.n_chunks = s->n_sorted / kChunkPts;
        ga.inv_chunks = 1.0f / (float)ga.n_chunks;
        /* 16 hypotheses per group. Groups of 32 (TM_GROUP_HYPS=32) let more near-identical poses meet in one
         * item and fetch fewer voxel lines (4.6e8 per batch against 5.2e8), but the larger item costs as much
         * as that saves: 13.05 ms against 12.90 ms in the same run. */
        ga.group_hyps = 16u;
        ga.n_groups = (n_hyp + ga.group_hyps - 1) / ga.group_hyps;
        const size_t n_slots_g = (size_t)ga.n_groups * ga.n_chunks;
        /* slots are addressed with 32 bits and the chunks ride in blockIdx.y: beyond that (billions of points,
         * or hundreds of millions of (group, chunk) slots in one call) the per-hypothesis kernel takes the launch */
        if (n_slots_g >= 0xFFFFFFFFull || (ga.n_chunks + 255) / 256 > 65535u) grouped = false;
        if (n_slots_g > slot_budget) grouped = false; /* one group alone exceeds the budget (>1e9-point scene) */
        if (grouped) {
            TM_FORWARD(s->w_masks.ensure(n_slots_g));
            TM_FORWARD(s->w_items.ensure(n_slots_g + 1));
            TM_FORWARD(s->w_aux.ensure(n_hyp));
            ga.aux = s->w_aux.p;
            hipLaunchKernelGGL(k_pose_aux, dim3((n_hyp + 255) / 256), dim3(256), 0, c->stream, d_poses, n_hyp,
                               s->coord_max, s->w_aux.p);
            ga.masks = s->w_masks.p;
            ga.items = s->w_items.p;
            ga.n_items = s->w_items.p + n_slots_g;
            d_n_items = ga.n_items;
            /* dynamic hand-out of the items: eight head words (one per XCD), 128 B apart */
            TM_FORWARD(s->w_heads.ensure(8 * 32));
            TM_HIP_CHECK(hipMemsetAsync(s->w_heads.p, 0, 8 * 32 * 4, c->stream));
            ga.heads = s->w_heads.p;
            hipLaunchKernelGGL(k_group_cull, dim3(ga.n_groups, ((ga.n_chunks + 7) / 8 + 255) / 256), dim3(256), 0, c->stream,
                               a, ca, ga);
            size_t tmp = 0;
            rocprim::counting_iterator<uint32_t> slots(0u);
            auto flags = rocprim::make_transform_iterator(s->w_masks.p, nonzero_mask());
            TM_HIP_CHECK(rocprim::select(nullptr, tmp, slots, flags, s->w_items.p, s->w_items.p + n_slots_g,
                                         n_slots_g, c->stream));
            TM_FORWARD(scratch_ensure(c, tmp));
            TM_HIP_CHECK(rocprim::select(c->scratch.p, tmp, slots, flags, s->w_items.p, s->w_items.p + n_slots_g,
                                         n_slots_g, c->stream));
            /* auto: on for large batches against a voxel table of 256 MB and more (the configs[1] model: 702 MB,
             * +5 %; the 100k-point model of config 5: +3 %). The unsorted order (group-major, chunks along the scene's
             * curve) already has model-space locality — consecutive items share a pose and cover neighbouring scene
             * chunks — and the unordered scatter gives that up inside a cell: with a table that fits the Infinity
             * Cache anyway the binning only costs (plane2 uvw model, 21 MB: 7.4 -> 8.0 ms; 23k-point model, 79 MB:
             * 4.4 -> 5.0 ms; the cylinder's 36 MB sheet is the exception, 12.4 -> 11.4 ms with item_sort = 4) */
            int item_sort = c->opt.item_sort;
            if (item_sort < 0) item_sort = (n_hyp >= 8192u && m->voxel_slots * 4ull >= (256ull << 20)) ? 4 : 0;
            if (item_sort) { /* the items binned by the model region they land in (see k_item_bins) */
                /* cells of 2^shift voxels, Morton-ordered */
                const uint32_t cell_shift = (uint32_t)std::max(1, item_sort);
                uint32_t max_ext = (uint32_t)std::max(m->extents[0], std::max(m->extents[1], m->extents[2]));
                uint32_t bits = 1;
                while ((1u << bits) < ((max_ext >> cell_shift) + 1u)) ++bits;
                const uint32_t n_bins = 1u << (3u * std::min(bits, 8u));
                TM_FORWARD(s->w_item_bin.ensure(n_slots_g));
                TM_FORWARD(s->w_items_sorted.ensure(n_slots_g + 1));
                TM_FORWARD(s->w_item_hist.ensure(2 * (size_t)n_bins + 2));
                uint32_t* d_hist = s->w_item_hist.p;
                uint32_t* d_offs = s->w_item_hist.p + n_bins + 1;
                TM_HIP_CHECK(hipMemsetAsync(d_hist, 0, ((size_t)n_bins + 1) * 4, c->stream));
                hipLaunchKernelGGL(k_item_bins, dim3(1024), dim3(256), 0, c->stream, a, ga, n_bins, cell_shift, s->w_item_bin.p,
                                   d_hist);
                size_t tmp2 = 0;
                TM_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp2, d_hist, d_offs, 0u, (size_t)n_bins + 1,
                                                     rocprim::plus<uint32_t>(), c->stream));
                TM_FORWARD(scratch_ensure(c, tmp2));
                TM_HIP_CHECK(rocprim::exclusive_scan(c->scratch.p, tmp2, d_hist, d_offs, 0u, (size_t)n_bins + 1,
                                                     rocprim::plus<uint32_t>(), c->stream));
                hipLaunchKernelGGL(k_item_scatter, dim3(1024), dim3(256), 0, c->stream, ga, s->w_item_bin.p, d_offs,
                                   s->w_items_sorted.p);
                ga.items = s->w_items_sorted.p;
            }
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
            /* persistent waves with a static share of the items each: the grid must be exactly what is
             * resident at once, or the surplus workgroups run their share after everyone else */
            /* every table a gather indexes below 4 GiB: 32-bit byte offsets (ld_off32) */
            const bool w32 = c->opt.score_off32 && m->voxel_slots < (1ull << 30) && s->n_sorted < (1u << 28) && m->n < (1u << 28);
            /* nobody reads the stage counters outside a measurement window with counters: the kernel does not keep them */
            const bool stats = c->timing_mode == 1 || c->opt.debug_stats != 0;
            auto kern = stats ? items_kernel<true>(l2, d_centers != nullptr, near_metric, w32)
                              : items_kernel<false>(l2, d_centers != nullptr, near_metric, w32);
            stats_kept = stats;
            int per_cu = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0);
            if (per_cu < 1) per_cu = 1;
            uint32_t wgs = (uint32_t)cus * (uint32_t)per_cu;
            if (c->opt.score_item_wgs > 0) wgs = (uint32_t)std::max(8, c->opt.score_item_wgs);
            wgs &= ~7u;
            TM_FORWARD(s->w_stats.ensure(std::max<size_t>(n_slots, wgs)));
            a.stat_slots = s->w_stats.p;
            n_stat_slots = wgs;
            if (c->opt.debug_stats)
                fprintf(stderr, "[tm] grouped: %u persistent workgroups (%d per CU)\n", wgs, per_cu);
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, c->stream, a, ca, ga);
        } else {
        /* split every hypothesis over three workgroups (rounds of 1024 tiles taken in turn): shorter
         * workgroups even out the tail of the launch */
        uint32_t parts = 3u;
        parts = std::min<uint32_t>(parts, std::max<uint32_t>(1u, (s->n_tiles + kCullChunk - 1) / kCullChunk));
        if (parts > 1) TM_HIP_CHECK(hipMemsetAsync(s->w_stats.p, 0, n_slots * sizeof(uint4), c->stream));
        const dim3 cgrid(n_hyp, parts);
        if (l2 && !d_active)
            hipLaunchKernelGGL((k_score_culled<true, false>), cgrid, dim3(256), 0, c->stream, a, ca);
        else if (l2)
            hipLaunchKernelGGL((k_score_culled<true, true>), cgrid, dim3(256), 0, c->stream, a, ca);
        else if (near_metric && !d_active)
            hipLaunchKernelGGL((k_score_culled<false, false, true>), cgrid, dim3(256), 0, c->stream, a, ca);
        else if (near_metric)
            hipLaunchKernelGGL((k_score_culled<false, true, true>), cgrid, dim3(256), 0, c->stream, a, ca);
        else if (!d_active)
            hipLaunchKernelGGL((k_score_culled<false, false>), cgrid, dim3(256), 0, c->stream, a, ca);
        else
            hipLaunchKernelGGL((k_score_culled<false, true>), cgrid, dim3(256), 0, c->stream, a, ca);
        }
    } else if (d_active) {
        hipLaunchKernelGGL(k_score_brute<true>, grid, dim3(256), 0, c->stream, a);
    } else {
        hipLaunchKernelGGL(k_score_brute<false>, grid, dim3(256), 0, c->stream, a);
    }
    TM_HIP_CHECK(hipGetLastError());
    if (c->timing) TM_HIP_CHECK(hipEventRecord(c->ev_ring[2 * trow + 1], c->stream));
    hipLaunchKernelGGL(k_finalize_scores, dim3((n_hyp + 255) / 256), dim3(256), 0, c->stream,
                       s->w_scores_fx.p, n_hyp, 1.0 / kFx, (double)m->n, d_scores);
    TM_HIP_CHECK(hipGetLastError());
    if (c->timing) { /* no host synchronisation here: the counters stay on the device until asked for */
        if (stats_kept) {
            hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(1024), 0, c->stream, s->w_stats.p, (uint32_t)n_stat_slots,
                               d_n_items,
                               c->stats.p + (size_t)trow * kTimedWords);
            TM_HIP_CHECK(hipGetLastError());
        } else { /* events only (tm_ctx_enable_timing(ctx, 2)): the row reads zero */
            TM_HIP_CHECK(hipMemsetAsync(c->stats.p + (size_t)trow * kTimedWords, 0, kTimedWords * 8, c->stream));
        }
        c->timed_hyp[trow] = n_hyp;
        c->timed_evals_host[trow] = culled ? ~0ull : (uint64_t)n_hyp * n_active;
        ++c->n_timed;
        if (c->opt.debug_stats) {
            score_timing t;
            uint32_t nl = 0;
            TM_FORWARD(read_timed(c, c->n_timed - 1, &t, &nl));
            fprintf(stderr, "[tm] score: evals %llu in_grid %llu near %llu passing %llu ms %.3f\n",
                    (unsigned long long)t.evaluations, (unsigned long long)t.in_grid, (unsigned long long)t.near,
                    (unsigned long long)t.passing, t.ms);
        }
    }
    return TM_OK;
}

static uint64_t count_active(const tm_scene* s) {
    uint64_t live = s->n_live;
    (void)tmx::scene_live_count(s, &live); /* fetched now if the last mask change left it on the device */
    return live;
}

static int generate_device(tm_scene* s, tm_model* m, const uint2* d_ij, uint32_t n_pairs, float minf,
                           float maxf, int force_up, int query_limit, float* d_poses, uint4* d_prov,
                           uint32_t cap, uint32_t* count, uint32_t* count_total) {
    TM_FORWARD(check_pair(s, m));
    if (!m->pairs_ready) {
        set_error("Cannot query uninitialized model");
        return TM_ERR_NOT_INITIALIZED;
    }
    if (count) *count = 0;
    if (count_total) *count_total = 0;
    if (n_pairs == 0) return TM_OK;
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    gen_args g;
    g.opos = s->opos.p;
    g.otan = s->otan.p;
    g.n_scene = s->n;
    g.mpos = m->pos.p;
    g.mtan = m->tan4.p;
    g.ht_slot = m->ht_slot.p;
    g.ht_key = m->ht_key.p;
    g.cap_mask = m->ht_cap - 1;
    g.seg_begin = m->seg_begin.p;
    g.entries = m->entries.p;
    g.new2orig = m->new2orig.p;
    float lower = m->diameter * minf, upper = m->diameter * maxf;
    g.lower2 = lower * lower;
    g.upper2 = upper * upper;
    g.lo0 = m->bmin[0];
    g.hi0 = m->bmax[0];
    g.steps = (uint32_t)m->distance_step_count;
    g.angle_step = m->angle_step;
    g.limit = query_limit > 0 ? (uint32_t)query_limit : 0xFFFFFFFFu; /* reference :310 */
    g.force_up = force_up;
    TM_FORWARD(s->w_u32a.ensure(n_pairs));          /* seg_first */
    TM_FORWARD(s->w_u32b.ensure((size_t)n_pairs + 1)); /* take, then offsets */
    TM_FORWARD(s->w_u32c.ensure((size_t)n_pairs + 1));
    uint32_t* d_first = s->w_u32a.p;
    uint32_t* d_take = s->w_u32b.p;
    uint32_t* d_offs = s->w_u32c.p;
    TM_HIP_CHECK(hipMemsetAsync(d_take + n_pairs, 0, 4, c->stream));
    hipLaunchKernelGGL(k_gen_lookup, dim3((n_pairs + 255) / 256), dim3(256), 0, c->stream, g, d_ij, n_pairs,
                       d_first, d_take);
    TM_HIP_CHECK(hipGetLastError());
    size_t tmp = 0;
    TM_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp, d_take, d_offs, 0u, (size_t)n_pairs + 1,
                                         rocprim::plus<uint32_t>(), c->stream));
    TM_FORWARD(scratch_ensure(c, tmp));
    TM_HIP_CHECK(rocprim::exclusive_scan(c->scratch.p, tmp, d_take, d_offs, 0u, (size_t)n_pairs + 1,
                                         rocprim::plus<uint32_t>(), c->stream));
    uint32_t total = 0;
    TM_HIP_CHECK(hipMemcpyAsync(&total, d_offs + n_pairs, 4, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (total == 0) return TM_OK;
    if (!force_up) {
        hipLaunchKernelGGL(k_gen_emit, dim3((total + 255) / 256), dim3(256), 0, c->stream, g, d_ij, n_pairs,
                           d_first, d_offs, total, cap, d_poses, d_prov, (uint32_t*)nullptr);
        TM_HIP_CHECK(hipGetLastError());
        if (count) *count = std::min(total, cap);
        if (count_total) *count_total = total;
        return TM_OK;
    }
    /* force_up: emit everything to a staging area, then compact the survivors in order */
    dbuf<float> st_pose;
    dbuf<uint4> st_prov;
    dbuf<uint32_t> keep, dst;
    int rc;
    auto cleanup = [&]() {
        st_pose.release();
        st_prov.release();
        keep.release();
        dst.release();
    };
    if ((rc = st_pose.alloc((size_t)total * 16)) || (rc = st_prov.alloc(total)) ||
        (rc = keep.alloc((size_t)total + 1)) || (rc = dst.alloc((size_t)total + 1))) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemsetAsync(keep.p + total, 0, 4, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gen_emit, dim3((total + 255) / 256), dim3(256), 0, c->stream, g, d_ij, n_pairs,
                           d_first, d_offs, total, total, st_pose.p, st_prov.p, keep.p);
        e = hipGetLastError();
    }
    size_t tmp2 = 0;
    if (e == hipSuccess)
        e = rocprim::exclusive_scan(nullptr, tmp2, keep.p, dst.p, 0u, (size_t)total + 1,
                                    rocprim::plus<uint32_t>(), c->stream);
    if (e == hipSuccess && scratch_ensure(c, tmp2) != TM_OK) {
        cleanup();
        return TM_ERR_HIP;
    }
    if (e == hipSuccess)
        e = rocprim::exclusive_scan(c->scratch.p, tmp2, keep.p, dst.p, 0u, (size_t)total + 1,
                                    rocprim::plus<uint32_t>(), c->stream);
    uint32_t kept = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&kept, dst.p + total, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_compact, dim3((total + 255) / 256), dim3(256), 0, c->stream,
                           (const float4*)st_pose.p, st_prov.p, keep.p, dst.p, total, cap, (float4*)d_poses,
                           d_prov);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) {
        set_error("tm_generate_hypotheses: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    if (count) *count = std::min(kept, cap);
    if (count_total) *count_total = kept;
    return TM_OK;
}

}  // namespace tmx

using namespace tmx;

extern "C" {

int tm_scene_create(tm_ctx* ctx, const float* surfels, uint32_t n, const int32_t* tangent_mask,
                    tm_scene** out) {
    TM_GUARD(ctx);
    TM_REQUIRE(ctx && surfels && out, TM_ERR_INVALID, "tm_scene_create: null argument");
    TM_REQUIRE(n > 0, TM_ERR_INVALID, "tm_scene_create: empty cloud");
    TM_HIP_CHECK(hipSetDevice(ctx->device));
    tm_scene* s = new tm_scene();
    s->ctx = ctx;
    s->n = n;
    const bool dbg = ctx->opt.debug_times != 0;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[scene create] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (tangent_mask) s->h_tmask.assign(tangent_mask, tangent_mask + n);
    else s->h_tmask.assign(n, 0); /* null: no tangent point yet (tm_scene_auto_tangent_mask / _set_tangent_mask follow) */
    s->h_mask.assign(n, 0);
    lap("host masks");
    /* no host mirror of the cloud here: the helpers that read rows on the host fetch it from the device copy when
     * they are first called (tmx::scene_mirror) */
    int rc = scene_upload(s, surfels);
    lap("upload");
    if (!rc) rc = scene_build(s, true);
    lap("build");
    if (rc) {
        tm_scene_destroy(s);
        return rc;
    }
    *out = s;
    return TM_OK;
}

int tm_scene_create_dev(tm_ctx* ctx, const float* d_surfels, uint32_t n, const int32_t* d_tangent_mask, tm_scene** out) {
    TM_GUARD(ctx);
    TM_REQUIRE(ctx && d_surfels && out, TM_ERR_INVALID, "tm_scene_create_dev: null argument");
    TM_REQUIRE(n > 0, TM_ERR_INVALID, "tm_scene_create_dev: empty cloud");
    TM_HIP_CHECK(hipSetDevice(ctx->device));
    tm_scene* s = new tm_scene();
    s->ctx = ctx;
    s->n = n;
    s->h_tmask.assign(n, 0);
    s->h_mask.assign(n, 0);
    int rc = scene_upload(s, nullptr); /* allocates d_cloud / d_tmask / d_mask and uploads the (zero) masks */
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(s->d_cloud.p, d_surfels, (size_t)n * 48, hipMemcpyDeviceToDevice, ctx->stream);
    if (!rc && e == hipSuccess && d_tangent_mask) {
        e = hipMemcpyAsync(s->d_tmask.p, d_tangent_mask, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) /* the host-side helpers read tangent_mask_ from their mirror */
            e = hipMemcpyAsync(s->h_tmask.data(), d_tangent_mask, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (!rc && e != hipSuccess) {
        set_error("tm_scene_create_dev: %s", hipGetErrorString(e));
        rc = TM_ERR_HIP;
    }
    if (!rc) rc = scene_build(s, true);
    if (rc) {
        tm_scene_destroy(s);
        return rc;
    }
    *out = s;
    return TM_OK;
}

int tm_scene_destroy(tm_scene* s) {
    if (!s) return TM_OK;
    if (!ctx_alive(s->ctx)) { /* context already gone: the device memory went with it */
        delete s;
        return TM_OK;
    }
    TM_GUARD(s);
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    tmx::scene_pinned_give(s->ctx, s->h_pinned, s->h_pinned_bytes);
    s->h_pinned = nullptr;
    s->d_cloud.release();
    s->d_tmask.release();
    s->d_tangent_idx.release();
    s->ed_tlist.release();
    s->ed_row_info.release();
    s->ed_sort_scratch.release();
    s->ed_plist.release();
    s->ed_plist_n.release();
    s->ed_plist_p.release();
    s->ed_blist.release();
    s->ed_blist_n.release();
    s->ed_hist.release();
    s->ed_fb.release();
    s->w_item_bin.release();
    s->w_items_sorted.release();
    s->w_item_hist.release();
    s->d_mask.release();
    s->d_order.release();
    s->d_keys.release();
    s->d_bounds.release();
    s->opos.release();
    s->otan.release();
    s->oref.release();
    s->spos.release();
    s->sref.release();
    s->spos3.release();
    s->tiles.release();
    s->supers.release();
    s->subs.release();
    s->chunks.release();
    s->w_masks.release();
    s->w_heads.release();
    s->w_aux.release();
    s->w_items.release();
    s->w_poses.release();
    s->w_counts.release();
    s->w_scores_fx.release();
    s->w_scores.release();
    s->w_u32a.release();
    s->w_u32b.release();
    s->w_u32c.release();
    s->w_stats.release();
    s->w_pairs.release();
    s->w_words.release();
    s->w_corr.release();
    s->w_owner.release();
    s->w_kept_tan.release();
    s->w_mask_idx.release();
    s->w_active_sorted.release();
    s->ed_subset.release();
    s->ed_dropped.release();
    s->ed_tests.release();
    s->ed_saved.release();
    s->ed_limit.release();
    s->ed_pos.release();
    s->ed_tail.release();
    s->ed_corr.release();
    s->ed_tilebits.release();
    s->ed_member.release();
    s->ed_slot.release();
    s->ed_centres.release();
    s->ed_wprefix.release();
    s->ed_S.release();
    s->ed_pending.release();
    s->sb_ij.release();
    s->sb_pair_slot.release();
    s->sb_centres.release();
    s->sb_prov.release();
    s->sb_result.release();
    s->w_active.release();
    s->ed_cpos.release();
    delete s;
    return TM_OK;
}

int tm_scene_set_mask(tm_scene* s, const int32_t* mask) {
    TM_GUARD(s);
    TM_REQUIRE(s, TM_ERR_INVALID, "null scene");
    TM_HIP_CHECK(hipSetDevice(s->ctx->device));
    if (mask) {
        s->h_mask.assign(mask, mask + s->n);
        TM_HIP_CHECK(hipMemcpyAsync(s->d_mask.p, static_cast<const tmx::mask_mirror&>(s->h_mask).data(), (size_t)s->n * 4,
                                    hipMemcpyHostToDevice, s->ctx->stream));
    } else {
        if (!s->h_mask_stale && s->h_mask.all_zero() && s->h_mask.size() == s->n && s->n_sorted) return TM_OK; /* clear already */
        /* the mirror is zeroed only if somebody wrote to it: a search that adds its matches on the device
         * (tm_scene_mask_add, _extracted) never does, and 4 MB of calloc per search were a tenth of a millisecond */
        if (!(s->h_mask.all_zero() && s->h_mask.size() == s->n)) s->h_mask.assign(s->n, 0);
        TM_HIP_CHECK(hipMemsetAsync(s->d_mask.p, 0, (size_t)s->n * 4, s->ctx->stream));
    }
    s->h_mask_stale = false;
    return scene_build(s, false);
}

/* mask_[idx] = 1 for the listed points (include/impl/scene.hpp:87-90: the accepted matches' correspondences), on
 * top of the mask the scene holds: the indices go up (4 bytes each) instead of the whole mask (4 bytes per scene
 * point, twice: host mirror and upload), the device sets its words and derives the masked arrays again. */
__global__ __launch_bounds__(256) void k_mask_set(const uint32_t* __restrict__ idx, uint32_t n, int32_t* __restrict__ mask) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) mask[idx[t]] = 1;
}
int tm_scene_mask_add(tm_scene* s, const uint32_t* indices, uint32_t n_idx) {
    TM_GUARD(s);
    TM_REQUIRE(s && (indices || n_idx == 0), TM_ERR_INVALID, "tm_scene_mask_add: null argument");
    if (n_idx == 0) return TM_OK;
    uint32_t mx = 0;
    for (uint32_t k = 0; k < n_idx; ++k) mx = std::max(mx, indices[k]);
    TM_REQUIRE(mx < s->n, TM_ERR_INVALID, "tm_scene_mask_add: index out of range");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    /* the host mirror of mask_ is not touched here (it would be one scattered write per index into a 4 n-byte array):
     * the helpers that read it (tm_pair_is_valid_sample, the live count) fetch it from the device when it is stale */
    s->h_mask_stale = true;
    TM_FORWARD(tmx::scene_pinned_ensure(s, (size_t)n_idx * 4));
    memcpy(s->h_pinned, indices, (size_t)n_idx * 4);
    TM_FORWARD(s->w_mask_idx.ensure(n_idx));
    TM_HIP_CHECK(hipMemcpyAsync(s->w_mask_idx.p, s->h_pinned, (size_t)n_idx * 4, hipMemcpyHostToDevice, c->stream));
    TM_HIP_CHECK(hipEventRecord(c->ev_copy, c->stream));
    hipLaunchKernelGGL(k_mask_set, dim3((n_idx + 255) / 256), dim3(256), 0, c->stream, s->w_mask_idx.p, n_idx, s->d_mask.p);
    TM_HIP_CHECK(hipGetLastError());
    const int rc = scene_build(s, false); /* queued, not waited for */
    TM_HIP_CHECK(hipEventSynchronize(c->ev_copy)); /* the staging buffer is free again; the kernels run on */
    return rc;
}

int tm_scene_set_tangent_mask(tm_scene* s, const int32_t* tangent_mask) {
    TM_GUARD(s);
    TM_REQUIRE(s && tangent_mask, TM_ERR_INVALID, "tm_scene_set_tangent_mask: null argument");
    TM_REQUIRE(!s->h_sparse, TM_ERR_UNSUPPORTED, "tm_scene_set_tangent_mask: not for a projected scene");
    TM_HIP_CHECK(hipSetDevice(s->ctx->device));
    s->h_tmask.assign(tangent_mask, tangent_mask + s->n);
    TM_HIP_CHECK(hipMemcpyAsync(s->d_tmask.p, s->h_tmask.data(), (size_t)s->n * 4, hipMemcpyHostToDevice, s->ctx->stream));
    s->tangent_idx_ready = false; /* the ascending list of tangent points is rebuilt on its next use */
    s->n_sorted = 0;              /* the refs of the sorted points depend on the mask: gather again, same order */
    return scene_build(s, false);
}

uint32_t tm_scene_size(const tm_scene* s) {
    TM_GUARD(s); return s ? s->n : 0; }
uint64_t tm_scene_live_count(const tm_scene* s) {
    TM_GUARD(s);
    return s ? count_active(s) : 0;
}

int tm_scene_download(const tm_scene* s, float* surfels, int32_t* tangent_mask) {
    TM_GUARD(s);
    TM_REQUIRE(s, TM_ERR_INVALID, "null scene");
    if (s->h_sparse && surfels) { /* a projected scene keeps only its tangent points on the host: fetch the rest */
        TM_HIP_CHECK(hipSetDevice(s->ctx->device));
        TM_HIP_CHECK(hipMemcpyAsync(surfels, s->d_cloud.p, (size_t)s->n * 48, hipMemcpyDeviceToHost, s->ctx->stream));
        TM_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));
    } else if (surfels) {
        TM_FORWARD(scene_mirror(s));
        memcpy(surfels, s->h_cloud.data(), (size_t)s->n * 48);
    }
    if (tangent_mask) memcpy(tangent_mask, s->h_tmask.data(), (size_t)s->n * 4);
    return TM_OK;
}

/* the points with tangent_mask set, ascending, with their surfels as the kernels see them (uvw surfels for a
 * projected scene): all the host side of a search reads of a scene's cloud */
int tm_scene_tangent_points(const tm_scene* s, uint32_t* indices, float* surfels, uint32_t cap, uint32_t* count) {
    TM_GUARD(s);
    TM_REQUIRE(s && count, TM_ERR_INVALID, "tm_scene_tangent_points: null argument");
    if (s->h_sparse) {
        *count = (uint32_t)s->h_tan_idx.size();
        if (!indices && !surfels) return TM_OK;
        TM_REQUIRE(cap >= *count, TM_ERR_CAPACITY, "tm_scene_tangent_points: buffers too small");
        if (indices) memcpy(indices, s->h_tan_idx.data(), (size_t)*count * 4);
        if (surfels) memcpy(surfels, s->h_tan_rows.data(), (size_t)*count * 48);
        return TM_OK;
    }
    uint32_t k = 0;
    if (surfels) TM_FORWARD(scene_mirror(s));
    for (uint32_t i = 0; i < s->n; ++i) {
        if (!s->h_tmask[i]) continue;
        if ((indices || surfels) && k < cap) {
            if (indices) indices[k] = i;
            if (surfels) memcpy(surfels + 12 * (size_t)k, s->h_cloud.data() + 12 * (size_t)i, 48);
        }
        ++k;
    }
    *count = k;
    TM_REQUIRE(!(indices || surfels) || cap >= k, TM_ERR_CAPACITY, "tm_scene_tangent_points: buffers too small");
    return TM_OK;
}

int tm_pair_valid_raw(const float* a, const float* b, float diameter, float minf, float maxf, const float bmin[4],
                      const float bmax[4]) {
    if (!a || !b || !bmin || !bmax) return 0;
    f3 p1 = make3(a[0], a[1], a[2]), p2 = make3(b[0], b[1], b[2]);
    f3 t1 = make3(a[9], a[10], a[11]), t2 = make3(b[9], b[10], b[11]);
    float lower = diameter * minf, upper = diameter * maxf;
    lower *= lower;
    upper *= upper;
    f3 d = p2 - p1;
    float sq = len2(d);
    d = unit(d);
    if (sq < lower || sq > upper) return 0;
    if (1.f - fabsf(dot3(d, t1)) < 0.01f) return 0;
    feat4 f = pair_feature(p1, t1, p2, t2);
    return feature_valid(f, bmin[0], bmax[0]) ? 1 : 0;
}

int tm_pair_is_valid_sample(const tm_scene* s, const tm_model* m, uint32_t i, uint32_t j, float minf,
                            float maxf) {
    TM_GUARD(s);
    if (!s || !m || !m->pairs_ready || i >= s->n || j >= s->n || i == j) return 0;
    if (tmx::scene_host_mask(s) != TM_OK) return TM_ERR_HIP;
    if (s->h_mask[i] || !s->h_tmask[j] || s->h_mask[j]) return 0;
    /* a sparse mirror holds the tangent points: i is a first sample, hence one of them. A full mirror is fetched from
     * the device on first use (48 bytes per point + a stream synchronise): if that copy fails the answer is an error
     * code (negative), not "invalid sample" — a search must not degrade silently */
    const float* ri = scene_row(s, i);
    const float* rj = scene_row(s, j);
    if (!s->h_sparse && (!ri || !rj)) return TM_ERR_HIP; /* scene_mirror has set the error text */
    return tm_pair_valid_raw(ri, rj, m->diameter, minf, maxf, m->bmin, m->bmax);
}

int tm_generate_hypotheses_dev(tm_scene* s, tm_model* m, const uint32_t* d_ij_pairs, uint32_t n_pairs,
                               float min_diameter_factor, float max_diameter_factor, int force_up,
                               int query_limit, float* d_poses16, uint32_t* d_provenance4, uint32_t cap,
                               uint32_t* count, uint32_t* count_total) {
    TM_GUARD(s);
    TM_REQUIRE(d_ij_pairs && d_poses16 && d_provenance4, TM_ERR_INVALID,
               "tm_generate_hypotheses_dev: null argument");
    return generate_device(s, m, (const uint2*)d_ij_pairs, n_pairs, min_diameter_factor,
                           max_diameter_factor, force_up, query_limit, d_poses16, (uint4*)d_provenance4, cap,
                           count, count_total);
}

int tm_generate_hypotheses(tm_scene* s, tm_model* m, const uint32_t* ij_pairs, uint32_t n_pairs,
                           float min_diameter_factor, float max_diameter_factor, int force_up,
                           int query_limit, float* poses16, uint32_t* provenance4, uint32_t cap,
                           uint32_t* count, uint32_t* count_total) {
    TM_GUARD(s);
    TM_FORWARD(check_pair(s, m));
    TM_REQUIRE(ij_pairs || n_pairs == 0, TM_ERR_INVALID, "tm_generate_hypotheses: null pairs");
    if (count) *count = 0;
    if (count_total) *count_total = 0;
    if (n_pairs == 0) return m->pairs_ready ? TM_OK : TM_ERR_NOT_INITIALIZED;
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    /* the scene's work buffers (no allocation per call) */
    int rc;
    const uint32_t dev_cap = std::max(cap, 1u);
    TM_FORWARD(s->sb_ij.ensure((size_t)n_pairs * 2));
    TM_FORWARD(s->w_poses.ensure((size_t)dev_cap * 16));
    TM_FORWARD(s->sb_prov.ensure(dev_cap));
    hipError_t e = hipMemcpyAsync(s->sb_ij.p, ij_pairs, (size_t)n_pairs * 8, hipMemcpyHostToDevice, c->stream);
    uint32_t got = 0, total = 0;
    rc = TM_OK;
    if (e == hipSuccess)
        rc = generate_device(s, m, reinterpret_cast<const uint2*>(s->sb_ij.p), n_pairs, min_diameter_factor,
                             max_diameter_factor, force_up, query_limit, s->w_poses.p, s->sb_prov.p, cap, &got, &total);
    if (e == hipSuccess && rc == TM_OK && got) {
        if (poses16)
            e = hipMemcpyAsync(poses16, s->w_poses.p, (size_t)got * 64, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && provenance4)
            e = hipMemcpyAsync(provenance4, s->sb_prov.p, (size_t)got * 16, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        set_error("tm_generate_hypotheses: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    if (rc) return rc;
    if (count) *count = got;
    if (count_total) *count_total = total;
    return TM_OK;
}

int tm_score_hypotheses_dev(tm_scene* s, tm_model* m, int traits_kind, const float* d_poses16,
                            uint32_t n_hyp, float dist_thres, int mode, uint32_t* d_counts,
                            double* d_scores) {
    TM_GUARD(s);
    TM_REQUIRE(d_poses16 && d_counts && d_scores, TM_ERR_INVALID, "tm_score_hypotheses_dev: null argument");
    TM_FORWARD(check_pair(s, m));
    return score_device(s, m, traits_kind, d_poses16, n_hyp, nullptr, count_active(s), dist_thres, mode,
                        d_counts, d_scores);
}

int tm_score_hypotheses(tm_scene* s, tm_model* m, int traits_kind, const float* poses16, uint32_t n_hyp,
                        const int32_t* subset, uint32_t n_subset, float dist_thres, int mode,
                        uint32_t* counts, double* scores) {
    TM_GUARD(s);
    TM_FORWARD(check_pair(s, m));
    TM_REQUIRE(poses16 || n_hyp == 0, TM_ERR_INVALID, "tm_score_hypotheses: null poses");
    TM_REQUIRE(counts && scores, TM_ERR_INVALID, "tm_score_hypotheses: null outputs");
    if (n_hyp == 0) return TM_OK;
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_FORWARD(s->w_poses.ensure((size_t)n_hyp * 16));
    TM_FORWARD(s->w_counts.ensure(n_hyp));
    TM_FORWARD(s->w_scores.ensure(n_hyp));
    TM_HIP_CHECK(hipMemcpyAsync(s->w_poses.p, poses16, (size_t)n_hyp * 64, hipMemcpyHostToDevice, c->stream));
    const unsigned char* d_active = nullptr;
    uint64_t n_active = 0;
    if (subset) {
        /* flags per point: the scene keeps the host array (zero between calls) and the device buffer, so a call
         * costs its own subset, not a pass over the scene */
        for (uint32_t k = 0; k < n_subset; ++k)
            TM_REQUIRE(subset[k] >= 0 && (uint32_t)subset[k] < s->n, TM_ERR_INVALID,
                       "tm_score_hypotheses: subset index out of range");
        if (s->h_act.size() != s->n) s->h_act.assign(s->n, 0);
        TM_FORWARD(tmx::scene_host_mask(s));
        std::vector<unsigned char>& act = s->h_act;
        for (uint32_t k = 0; k < n_subset; ++k) {
            const uint32_t i = (uint32_t)subset[k];
            if (act[i]) continue;
            act[i] = 1;
            const float* row = scene_row(s, i); /* (a statistics figure; a sparse mirror counts unknown points as live) */
            if (!s->h_mask[i] && (!row || fin3(row))) ++n_active;
        }
        int rc0 = s->w_active.ensure(s->n);
        hipError_t e = rc0 ? hipErrorOutOfMemory
                           : hipMemcpyAsync(s->w_active.p, act.data(), s->n, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        for (uint32_t k = 0; k < n_subset; ++k) act[(uint32_t)subset[k]] = 0;
        if (e != hipSuccess) {
            set_error("tm_score_hypotheses: %s", hipGetErrorString(e));
            return rc0 ? rc0 : TM_ERR_HIP;
        }
        d_active = s->w_active.p;
    } else {
        n_active = count_active(s);
    }
    int rc = score_device(s, m, traits_kind, s->w_poses.p, n_hyp, d_active, n_active, dist_thres, mode,
                          s->w_counts.p, s->w_scores.p);
    hipError_t e = hipSuccess;
    if (rc == TM_OK) {
        e = hipMemcpyAsync(counts, s->w_counts.p, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(scores, s->w_scores.p, (size_t)n_hyp * 8, hipMemcpyDeviceToHost, c->stream);
    }
    hipError_t e2 = hipStreamSynchronize(c->stream);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) {
        set_error("tm_score_hypotheses: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

int tm_score_hypotheses_ball_dev(tm_scene* s, tm_model* m, const float* d_poses16, uint32_t n_hyp,
                                 const uint32_t* d_centers, float radius, float dist_thres, uint32_t* d_counts,
                                 double* d_scores) {
    TM_GUARD(s);
    TM_REQUIRE(d_poses16 && d_centers && d_counts && d_scores, TM_ERR_INVALID,
               "tm_score_hypotheses_ball_dev: null argument");
    TM_FORWARD(check_pair(s, m));
    /* any metric: the ball is the reference's radius_search_inclusive on the cloud the scene holds (uvw
     * points for a projected scene); the non-Euclidean metrics take the sphere-culled instantiation */
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    TM_REQUIRE(radius >= 0.f, TM_ERR_INVALID, "negative radius");
    return score_device(s, m, m->metric_kind, d_poses16, n_hyp, nullptr, count_active(s), dist_thres,
                        TM_SCORE_CULLED, d_counts, d_scores, d_centers, radius);
}

int tm_score_hypotheses_ball(tm_scene* s, tm_model* m, const float* poses16, uint32_t n_hyp,
                             const uint32_t* centers, float radius, float dist_thres, uint32_t* counts,
                             double* scores) {
    TM_GUARD(s);
    TM_FORWARD(check_pair(s, m));
    TM_REQUIRE(counts && scores, TM_ERR_INVALID, "tm_score_hypotheses_ball: null outputs");
    if (n_hyp == 0) return TM_OK;
    TM_REQUIRE(poses16 && centers, TM_ERR_INVALID, "tm_score_hypotheses_ball: null inputs");
    for (uint32_t h = 0; h < n_hyp; ++h)
        TM_REQUIRE(centers[h] < s->n, TM_ERR_INVALID, "tm_score_hypotheses_ball: centre index out of range");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_FORWARD(s->w_poses.ensure((size_t)n_hyp * 16));
    TM_FORWARD(s->w_counts.ensure(n_hyp));
    TM_FORWARD(s->w_scores.ensure(n_hyp));
    TM_FORWARD(s->w_u32a.ensure(n_hyp));
    TM_HIP_CHECK(hipMemcpyAsync(s->w_poses.p, poses16, (size_t)n_hyp * 64, hipMemcpyHostToDevice, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(s->w_u32a.p, centers, (size_t)n_hyp * 4, hipMemcpyHostToDevice, c->stream));
    TM_FORWARD(tm_score_hypotheses_ball_dev(s, m, s->w_poses.p, n_hyp, s->w_u32a.p, radius, dist_thres,
                                            s->w_counts.p, s->w_scores.p));
    TM_HIP_CHECK(hipMemcpyAsync(counts, s->w_counts.p, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(scores, s->w_scores.p, (size_t)n_hyp * 8, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return TM_OK;
}

/* finish_find for n poses in ONE set of launches (the poses side by side in the grid's y): per pose k count and score
 * sum -> d_count_fx[2 k], [2 k + 1]; correspondences (ascending scene index) -> d_corr + 2 * s->n * k (scene indices,
 * then model indices at + s->n). Ten dispatches per pose became seven per call: a search extracts a dozen poses. */
static int extract_batch_on_device(tm_scene* s, tm_model* m, const float* d_poses, uint32_t n, float dist_thres,
                                   unsigned long long* d_count_fx, uint32_t* d_corr) {
    tm_ctx* c = s->ctx;
    const uint32_t n_blocks = (s->n + 255) / 256, n_waves = n_blocks * 4;
    const size_t rows = (size_t)n * (n_waves + 1u);
    TM_FORWARD(s->w_u32a.ensure((size_t)s->n * n));
    TM_FORWARD(s->w_u32b.ensure(rows + 1));
    TM_FORWARD(s->w_u32c.ensure(rows + 1));
    TM_HIP_CHECK(hipMemsetAsync(d_count_fx, 0, (size_t)n * 16, c->stream));
    TM_HIP_CHECK(hipMemsetAsync(s->w_u32a.p, 0xFF, (size_t)s->n * n * 4, c->stream));
    score_args a;
    a.spos = s->spos.p;
    a.spos3 = s->spos3.p;
    a.sref = s->sref.p;
    a.mpos = m->pos.p;
    a.mpos3 = m->pos3.p;
    a.mref = m->ref.p;
    a.vox = m->voxels.p;
    a.active = nullptr;
    a.poses = d_poses;
    a.counts = nullptr;
    a.scores_fx = nullptr;
    a.g = m->grid;
    a.d2max = sq_threshold(dist_thres * m->resolution);
    a.n_hyp = n;
    a.hyp_per_block = 1;
    a.ablate = 0;
    a.kind = m->metric_kind;
    a.radius = m->metric_radius;
    a.dmax = dist_thres * m->resolution;
    a.stat_slots = nullptr;
    a.bcode = nullptr; /* no brick filter here; bx/by still address the bricked voxel table */
    a.bcode_tile = nullptr;
    a.bx = m->bdim[0];
    a.by = m->bdim[1];
    a.code_max = 255u;
    hipLaunchKernelGGL(k_extract_batch, dim3(s->n_sorted / 256, n), dim3(256), 0, c->stream, a, s->n_sorted, s->n, s->w_u32a.p,
                       d_count_fx, (const float4*)s->tiles.p, s->coord_max);
    /* ordered compaction on the device: only the correspondences travel back */
    hipLaunchKernelGGL(k_extract_count_batch, dim3(n_blocks, n), dim3(256), 0, c->stream, s->w_u32a.p, s->n, n_waves,
                       s->w_u32b.p);
    TM_HIP_CHECK(hipGetLastError());
    size_t tmp = 0;
    TM_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp, s->w_u32b.p, s->w_u32c.p, 0u, rows, rocprim::plus<uint32_t>(), c->stream));
    TM_FORWARD(scratch_ensure(c, tmp));
    TM_HIP_CHECK(rocprim::exclusive_scan(c->scratch.p, tmp, s->w_u32b.p, s->w_u32c.p, 0u, rows, rocprim::plus<uint32_t>(),
                                         c->stream));
    hipLaunchKernelGGL(k_extract_scatter_batch, dim3(n_blocks, n), dim3(256), 0, c->stream, s->w_u32a.p, s->n, n_waves,
                       s->w_u32c.p, m->new2orig.p, d_corr);
    TM_HIP_CHECK(hipGetLastError());
    return TM_OK;
}

/* finish_find (include/impl/scene.hpp:100-106) for n poses in one call: the passes are queued one after the other,
 * the counts come back in one read-back, the lists in a second one through pinned memory (the reference calls
 * finish_find once per candidate, :361-364 and at the end of icp_, :401). scene_corrs / model_corrs = n rows of
 * `stride` entries; counts[k] entries of row k are valid. TM_ERR_CAPACITY (counts valid) if stride is too small. */
/* The arbitration of find_parallel (include/impl/scene.hpp:193-237) over lists that are still on the device: candidate
 * `me` (1-based, in arbitration order) claims the scene points of its list that no ACCEPTED earlier candidate holds —
 * owner[idx] = number of the last candidate that claimed idx; whether that one was accepted follows from its two
 * counts (kept >= min_points_all and tangent points among them >= min_points), final since its launch has ended — and
 * counts them. One launch per candidate, in order; nothing is given back by hand (search.hpp arbitrate_marks is the
 * same rule on the host). */
__global__ __launch_bounds__(256) void k_arb_claim(const uint32_t* __restrict__ list, uint32_t count, uint32_t me,
                                                   unsigned char* __restrict__ owner, uint32_t* __restrict__ kept_tan,
                                                   const int32_t* __restrict__ tmask, uint32_t min_points,
                                                   uint32_t min_points_all) {
    __shared__ uint32_t s_k[4], s_t[4];
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    bool keep = false, tan = false;
    if (t < count) {
        const uint32_t idx = list[t];
        const uint32_t o = owner[idx];
        const bool held = o != 0u && kept_tan[2u * o] >= min_points_all && kept_tan[2u * o + 1u] >= min_points;
        if (!held) {
            owner[idx] = (unsigned char)me;
            keep = true;
            tan = tmask[idx] != 0;
        }
    }
    const uint32_t ck = (uint32_t)__popcll(__ballot(keep)), ct = (uint32_t)__popcll(__ballot(tan));
    if ((threadIdx.x & 63u) == 0u) s_k[threadIdx.x >> 6] = ck, s_t[threadIdx.x >> 6] = ct;
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t k = (s_k[0] + s_k[1]) + (s_k[2] + s_k[3]), tt = (s_t[0] + s_t[1]) + (s_t[2] + s_t[3]);
        if (k) atomicAdd(&kept_tan[2u * me], k);
        if (tt) atomicAdd(&kept_tan[2u * me + 1u], tt);
    }
}

/* finish_find of several poses; with `arb` also the arbitration of the candidates these poses are (see k_arb_claim):
 * verdict[k] = 1 accepted / 0 rejected, or -1 for every k when the decision needs the host (an accepted candidate
 * would have lost points to an earlier one and must be compacted; more than 254 candidates; a projected scene) */
struct arb_request {
    uint32_t min_points, min_points_all;
    int32_t* verdict;
    uint32_t *kept, *tangent_kept; /* may be null */
};
static int extract_batch_impl(tm_scene* s, tm_model* m, const float* poses16, uint32_t n, float dist_thres,
                              uint32_t* scene_corrs, uint32_t* model_corrs, uint32_t stride, uint32_t* counts,
                              double* scores, const arb_request* arb) {
    TM_FORWARD(check_pair(s, m));
    if (n == 0) return TM_OK;
    TM_REQUIRE(poses16 && counts, TM_ERR_INVALID, "tm_extract_correspondences_batch: null argument");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    TM_REQUIRE(n <= 4096u, TM_ERR_INVALID, "tm_extract_correspondences_batch: more than 4096 poses");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_FORWARD(s->w_poses.ensure((size_t)n * 16));
    TM_FORWARD(s->w_scores_fx.ensure((size_t)n * 2));
    TM_FORWARD(s->w_corr.ensure(2 * (size_t)s->n * n));
    TM_HIP_CHECK(hipMemcpyAsync(s->w_poses.p, poses16, (size_t)n * 64, hipMemcpyHostToDevice, c->stream));
    TM_FORWARD(extract_batch_on_device(s, m, s->w_poses.p, n, dist_thres, s->w_scores_fx.p, s->w_corr.p));
    std::vector<unsigned long long> h((size_t)n * 2);
    TM_HIP_CHECK(hipMemcpyAsync(h.data(), s->w_scores_fx.p, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    size_t total = 0;
    uint32_t mx = 0;
    for (uint32_t k = 0; k < n; ++k) {
        counts[k] = (uint32_t)h[2 * (size_t)k];
        if (scores) scores[k] = ((double)h[2 * (size_t)k + 1] * (1.0 / kFx)) / (double)m->n;
        total += counts[k];
        mx = std::max(mx, counts[k]);
    }
    s->extract_counts.assign(counts, counts + n); /* rows of w_corr a later tm_scene_mask_add_extracted may name */
    const bool lists = (scene_corrs || model_corrs) && total != 0;
    if (arb) {
        for (uint32_t k = 0; k < n; ++k) arb->verdict[k] = -1;
    }
    /* before anything else is queued: the caller comes back with room for the longest list */
    if (lists) TM_REQUIRE(stride >= mx, TM_ERR_CAPACITY, "tm_extract_correspondences_batch: stride smaller than a list");
    /* the arbitration, queued behind the lists: order = counts descending, stable (:193-195) */
    std::vector<uint32_t> order, h_kt;
    const bool decide = arb != nullptr && n <= 254u && s->d_tmask.p != nullptr;
    if (decide) {
        order.resize(n);
        for (uint32_t k = 0; k < n; ++k) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return counts[x] > counts[y]; });
        TM_FORWARD(s->w_owner.ensure(s->n));
        TM_FORWARD(s->w_kept_tan.ensure(2 * ((size_t)n + 1)));
        TM_HIP_CHECK(hipMemsetAsync(s->w_owner.p, 0, s->n, c->stream));
        TM_HIP_CHECK(hipMemsetAsync(s->w_kept_tan.p, 0, 2 * ((size_t)n + 1) * 4, c->stream));
        for (uint32_t r = 0; r < n; ++r) {
            const uint32_t k = order[r];
            if (!counts[k]) continue;
            hipLaunchKernelGGL(k_arb_claim, dim3((counts[k] + 255) / 256), dim3(256), 0, c->stream,
                               s->w_corr.p + 2 * (size_t)s->n * k, counts[k], r + 1u, s->w_owner.p, s->w_kept_tan.p,
                               s->d_tmask.p, arb->min_points, arb->min_points_all);
        }
        TM_HIP_CHECK(hipGetLastError());
        h_kt.resize(2 * ((size_t)n + 1));
        TM_HIP_CHECK(hipMemcpyAsync(h_kt.data(), s->w_kept_tan.p, h_kt.size() * 4, hipMemcpyDeviceToHost, c->stream));
    }
    uint32_t* stage = nullptr;
    if (lists) {
        const size_t need = total * 8;
        TM_FORWARD(tmx::scene_pinned_ensure(s, need));
        stage = (uint32_t*)s->h_pinned;
        size_t off = 0;
        for (uint32_t k = 0; k < n; ++k) {
            if (!counts[k]) continue;
            TM_HIP_CHECK(hipMemcpyAsync(stage + off, s->w_corr.p + 2 * (size_t)s->n * k, (size_t)counts[k] * 4,
                                        hipMemcpyDeviceToHost, c->stream));
            TM_HIP_CHECK(hipMemcpyAsync(stage + off + counts[k], s->w_corr.p + 2 * (size_t)s->n * k + s->n, (size_t)counts[k] * 4,
                                        hipMemcpyDeviceToHost, c->stream));
            off += 2 * (size_t)counts[k];
        }
    }
    if (lists || decide) TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (decide) {
        bool compaction = false;
        for (uint32_t r = 0; r < n; ++r) {
            const uint32_t k = order[r], kept = h_kt[2 * ((size_t)r + 1)], tan = h_kt[2 * ((size_t)r + 1) + 1];
            const bool acc = tan >= arb->min_points && kept >= arb->min_points_all;
            arb->verdict[k] = acc ? 1 : 0;
            if (arb->kept) arb->kept[k] = kept;
            if (arb->tangent_kept) arb->tangent_kept[k] = tan;
            compaction = compaction || (acc && kept != counts[k]);
        }
        if (compaction)
            for (uint32_t k = 0; k < n; ++k) arb->verdict[k] = -1; /* the host's arbitration trims the lists */
    }
    if (lists) {
        size_t off = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const bool wanted = !arb || arb->verdict[k] != 0; /* a rejected candidate's lists are not copied out */
            if (wanted && scene_corrs) memcpy(scene_corrs + (size_t)k * stride, stage + off, (size_t)counts[k] * 4);
            if (wanted && model_corrs) memcpy(model_corrs + (size_t)k * stride, stage + off + counts[k], (size_t)counts[k] * 4);
            off += 2 * (size_t)counts[k];
        }
    }
    return TM_OK;
}

int tm_extract_correspondences_batch(tm_scene* s, tm_model* m, const float* poses16, uint32_t n, float dist_thres,
                                     uint32_t* scene_corrs, uint32_t* model_corrs, uint32_t stride, uint32_t* counts,
                                     double* scores) {
    TM_GUARD(s);
    return extract_batch_impl(s, m, poses16, n, dist_thres, scene_corrs, model_corrs, stride, counts, scores, nullptr);
}

int tm_extract_arbitrate_batch(tm_scene* s, tm_model* m, const float* poses16, uint32_t n, float dist_thres,
                               uint32_t min_points, uint32_t min_points_all, uint32_t* scene_corrs, uint32_t* model_corrs,
                               uint32_t stride, uint32_t* counts, double* scores, int32_t* verdict, uint32_t* kept,
                               uint32_t* tangent_kept) {
    TM_GUARD(s);
    TM_REQUIRE(verdict || n == 0, TM_ERR_INVALID, "tm_extract_arbitrate_batch: null argument");
    arb_request arb{min_points, min_points_all, verdict, kept, tangent_kept};
    return extract_batch_impl(s, m, poses16, n, dist_thres, scene_corrs, model_corrs, stride, counts, scores, &arb);
}

/* mask_ |= the scene points of rows `rows` of the last tm_extract_*_batch call on this scene (their lists are still on
 * the device): what find_all does with the accepted matches (:87-90), without the lists travelling back up */
int tm_scene_mask_add_extracted(tm_scene* s, const uint32_t* rows, uint32_t n_rows) {
    TM_GUARD(s);
    TM_REQUIRE(s && (rows || n_rows == 0), TM_ERR_INVALID, "tm_scene_mask_add_extracted: null argument");
    if (n_rows == 0) return TM_OK;
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    for (uint32_t r = 0; r < n_rows; ++r)
        TM_REQUIRE(rows[r] < s->extract_counts.size(), TM_ERR_INVALID, "tm_scene_mask_add_extracted: no such row in the last extraction");
    s->h_mask_stale = true;
    bool any = false;
    for (uint32_t r = 0; r < n_rows; ++r) {
        const uint32_t cnt = s->extract_counts[rows[r]];
        if (!cnt) continue;
        hipLaunchKernelGGL(k_mask_set, dim3((cnt + 255) / 256), dim3(256), 0, c->stream,
                           s->w_corr.p + 2 * (size_t)s->n * rows[r], cnt, s->d_mask.p);
        any = true;
    }
    TM_HIP_CHECK(hipGetLastError());
    return any ? scene_build(s, false) : TM_OK;
}

int tm_extract_correspondences(tm_scene* s, tm_model* m, const float pose16[16], float dist_thres,
                               uint32_t* scene_corrs, uint32_t* model_corrs, uint32_t cap, uint32_t* count,
                               double* score) {
    TM_GUARD(s);
    TM_REQUIRE(pose16 && count, TM_ERR_INVALID, "tm_extract_correspondences: null argument");
    return tm_extract_correspondences_batch(s, m, pose16, 1, dist_thres, scene_corrs, model_corrs, cap, count, score);
}

static void icp_score_args(const tm_scene* s, const tm_model* m, float dist_thres, score_args& a) {
    a.spos = s->spos.p;
    a.spos3 = s->spos3.p;
    a.sref = s->sref.p;
    a.mpos = m->pos.p;
    a.mpos3 = m->pos3.p;
    a.mref = m->ref.p;
    a.vox = m->voxels.p;
    a.active = nullptr;
    a.poses = s->w_poses.p;
    a.counts = s->w_counts.p;
    a.scores_fx = s->w_scores_fx.p + 15;
    a.g = m->grid;
    a.d2max = sq_threshold(dist_thres * m->resolution);
    a.n_hyp = 1;
    a.hyp_per_block = 1;
    a.ablate = 0;
    a.kind = m->metric_kind;
    a.radius = m->metric_radius;
    a.dmax = dist_thres * m->resolution;
    a.stat_slots = nullptr;
    a.bcode = nullptr;
    a.bcode_tile = nullptr;
    a.bx = m->bdim[0];
    a.by = m->bdim[1];
    a.code_max = 255u;
}

int tm_icp_accumulate(tm_scene* s, tm_model* m, const float pose16[16], float dist_thres, uint32_t* count,
                      long long sums15[15]) {
    TM_GUARD(s);
    TM_FORWARD(check_pair(s, m));
    TM_REQUIRE(pose16 && count && sums15, TM_ERR_INVALID, "tm_icp_accumulate: null argument");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_FORWARD(s->w_poses.ensure(16));
    TM_FORWARD(s->w_counts.ensure(1));
    const uint32_t icp_blocks = std::min<uint32_t>((uint32_t)kIcpBlocks, s->n_sorted / 256);
    TM_FORWARD(s->w_scores_fx.ensure(16 + (size_t)kIcpBlocks * 16));
    TM_HIP_CHECK(hipMemcpyAsync(s->w_poses.p, pose16, 64, hipMemcpyHostToDevice, c->stream));
    score_args a;
    icp_score_args(s, m, dist_thres, a);
    hipLaunchKernelGGL(k_icp_accumulate, dim3(icp_blocks), dim3(256), 0, c->stream, a, s->n_sorted,
                       s->w_scores_fx.p + 16, (const icp_state*)nullptr, (const float4*)s->tiles.p, s->coord_max);
    hipLaunchKernelGGL(k_icp_finalize, dim3(1), dim3(kIcpFinalizeThreads), 0, c->stream, s->w_scores_fx.p + 16, icp_blocks,
                       s->w_scores_fx.p, 0);
    TM_HIP_CHECK(hipGetLastError());
    unsigned long long h_sums[16];
    TM_HIP_CHECK(hipMemcpyAsync(h_sums, s->w_scores_fx.p, 16 * 8, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    *count = (uint32_t)h_sums[15];
    for (int k = 0; k < 15; ++k) sums15[k] = (long long)h_sums[k];
    return TM_OK;
}

/* icp_ as one call (reference include/impl/scene.hpp:369-400 without the final finish_find): the passes, the
 * rigid fits and the "stop when the correspondences get fewer" test all run on the device, queued up front; one
 * read-back at the end. Same poses as the host loop over tm_icp_accumulate + search::detail::rigid_fit.
 * The batch form refines n independent candidates side by side (the reference refines the candidates of a round
 * one after the other, :169-171; they do not depend on one another): every pass is ONE accumulate launch over
 * (scene blocks, candidates) and ONE finalize launch with a workgroup per candidate, so a round's chain of
 * launches is max_iterations + 1 passes long instead of candidates x (max_iterations + 1). */
int tm_icp_refine_batch(tm_scene* s, tm_model* m, const float* poses16, uint32_t n, float dist_thres,
                        uint32_t max_iterations, float* out_poses16, uint32_t* out_counts, uint32_t* out_iterations) {
    TM_GUARD(s);
    TM_FORWARD(check_pair(s, m));
    if (n == 0) return TM_OK;
    TM_REQUIRE(poses16 && out_poses16 && out_counts, TM_ERR_INVALID, "tm_icp_refine_batch: null argument");
    TM_REQUIRE(n <= 65535u, TM_ERR_INVALID, "tm_icp_refine_batch: more than 65,535 candidates");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_FORWARD(s->w_poses.ensure((size_t)n * 16));
    TM_FORWARD(s->w_counts.ensure(1));
    const uint32_t icp_blocks = std::min<uint32_t>((uint32_t)kIcpBlocks, s->n_sorted / 256);
    const size_t state_words = (sizeof(icp_state) + 7) / 8;
    const size_t sums_words = (size_t)n * 16, part_words = (size_t)n * icp_blocks * 16;
    TM_FORWARD(s->w_scores_fx.ensure(sums_words + part_words + state_words * n));
    unsigned long long* sums = s->w_scores_fx.p;
    unsigned long long* partials = s->w_scores_fx.p + sums_words;
    icp_state* st = reinterpret_cast<icp_state*>(s->w_scores_fx.p + sums_words + part_words);
    TM_HIP_CHECK(hipMemcpyAsync(s->w_poses.p, poses16, (size_t)n * 64, hipMemcpyHostToDevice, c->stream));
    score_args a;
    icp_score_args(s, m, dist_thres, a);
    for (uint32_t pass = 0; pass <= max_iterations; ++pass) {
        hipLaunchKernelGGL(k_icp_accumulate, dim3(icp_blocks, n), dim3(256), 0, c->stream, a, s->n_sorted, partials,
                           pass ? st : (const icp_state*)nullptr, (const float4*)s->tiles.p, s->coord_max);
        hipLaunchKernelGGL(k_icp_finalize, dim3(n), dim3(kIcpFinalizeThreads), 0, c->stream, partials, icp_blocks, sums, pass ? 1 : 0,
                           s->w_poses.p, st, max_iterations, pass == 0 ? 1 : 0);
    }
    TM_HIP_CHECK(hipGetLastError());
    std::vector<icp_state> h(n);
    TM_HIP_CHECK(hipMemcpyAsync(h.data(), st, sizeof(icp_state) * n, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (uint32_t k = 0; k < n; ++k) {
        for (int e = 0; e < 16; ++e) out_poses16[(size_t)k * 16 + e] = h[k].best_t[e];
        out_counts[k] = (uint32_t)h[k].best_sums[15];
        if (out_iterations) out_iterations[k] = h[k].iter;
    }
    return TM_OK;
}

int tm_icp_refine(tm_scene* s, tm_model* m, const float pose16[16], float dist_thres, uint32_t max_iterations,
                  float out_pose16[16], uint32_t* out_count, uint32_t* out_iterations) {
    TM_GUARD(s);
    TM_REQUIRE(pose16 && out_pose16 && out_count, TM_ERR_INVALID, "tm_icp_refine: null argument");
    return tm_icp_refine_batch(s, m, pose16, 1, dist_thres, max_iterations, out_pose16, out_count, out_iterations);
}

/* the rigid fit of the ICP loop on the host (the same function the device loop runs) */
int tm_rigid_fit(const long long sums15[15], uint32_t count, float pose16[16]) {
    TM_REQUIRE(sums15 && pose16, TM_ERR_INVALID, "tm_rigid_fit: null argument");
    TM_REQUIRE(count >= 3, TM_ERR_INVALID, "tm_rigid_fit: fewer than three correspondences");
    rigid_fit_sums(sums15, count, pose16);
    return TM_OK;
}

int tm_best_of_batch(tm_ctx* c, const uint32_t* d_counts, const double* d_scores, const float* d_poses16,
                     uint32_t n_hyp, uint32_t id_offset, uint32_t id_stride, tm_best_record* best) {
    TM_GUARD(c);
    TM_REQUIRE(c && best, TM_ERR_INVALID, "tm_best_of_batch: null argument");
    TM_REQUIRE((d_counts && d_scores && d_poses16) || n_hyp == 0, TM_ERR_INVALID,
               "tm_best_of_batch: null device buffers");
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_FORWARD(scratch_ensure(c, sizeof(tm_best_record)));
    hipLaunchKernelGGL(k_best, dim3(1), dim3(1024), 0, c->stream, d_counts, d_scores, d_poses16, n_hyp,
                       id_offset, id_stride, (tm_best_record*)c->scratch.p);
    TM_HIP_CHECK(hipGetLastError());
    TM_HIP_CHECK(hipMemcpyAsync(best, c->scratch.p, sizeof(tm_best_record), hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return TM_OK;
}

}  // extern "C"
