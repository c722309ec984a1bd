/*
 * tm_model.hip — model side of the hot path: voxel NN grid (a6), pair enumeration + key table (a7),
 * query (a8) and voxel_query (a9). Replaces model<Point>::impl (reference include/impl/model.hpp).
 *
 * HBM layout: pos[n] float4 (xyz, w = 1 if |tangent| > 0.7), ref[n] float4 (tangent or normal),
 * tan4[n] float4 (raw tangent), all in Morton order; voxels[bricks*64] u32 (4x4x4-bricked); the pair table as a
 * CSR sorted by key: seg_key[n_keys] (packed key), seg_begin[n_keys+1], entries[] (i,j), plus an
 * open-addressing index keyed by MurmurHash3(key, seed 42) for the device-side probe.
 */
#include <cstring>

#include "tm_internal.h"
#include "tm_traits_math.h"
#include "tm_table.h"
#include "tm_knn.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <limits>

namespace tmx {

static inline bool fin3(const float* p) {
    return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
}

static inline uint64_t spread21(uint64_t v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

/* ------------------------------------------------------------------------------------------------
 * a6 — voxel NN grid. One thread owns VPT voxels; the cloud streams through LDS in tiles and every
 * thread scans the tile (broadcast reads), keeping the strict minimum => lowest index on ties.
 * Compute-bound brute force: 11 VALU ops per (voxel, point).
 * ---------------------------------------------------------------------------------------------- */
constexpr int kVoxTile = 2048; /* points per LDS tile: 32 KiB */
constexpr int kVoxVPT = 4;     /* voxels per thread */

/* Voxels are enumerated (and stored) brick-major: slot = brick*64 + (z&3)*16 + (y&3)*4 + (x&3), so a
 * wave owns exactly one 4x4x4 brick: coalesced 256-B stores and the brick's minimum distance falls
 * out of one wave reduction. `pos` is the Morton-ordered cloud with the ORIGINAL index in .w: ties in
 * d2 go to the lowest original index (what a scan in original order with a strict '<' would keep). */
__global__ __launch_bounds__(256) void k_voxel_nn(const float4* __restrict__ pos, uint32_t n,
                                                  grid_t g, uint64_t voxel_slots, int bx, int by,
                                                  uint32_t* __restrict__ voxels,
                                                  uint32_t* __restrict__ brick_min_bits) {
    __shared__ float4 tile[kVoxTile];
    float qx[kVoxVPT], qy[kVoxVPT], qz[kVoxVPT], best[kVoxVPT];
    uint32_t besti[kVoxVPT], besto[kVoxVPT];
    bool live[kVoxVPT];
    const uint64_t base = (uint64_t)blockIdx.x * (256 * kVoxVPT);
    const float isx = 1.0f / g.sx, isy = 1.0f / g.sy, isz = 1.0f / g.sz;
    const float itx = -(g.tx * isx), ity = -(g.ty * isy), itz = -(g.tz * isz);
#pragma unroll
    for (int v = 0; v < kVoxVPT; ++v) {
        uint64_t slot = base + (uint64_t)v * 256 + threadIdx.x;
        uint32_t brick = (uint32_t)(slot >> 6), loc = (uint32_t)(slot & 63);
        uint32_t bi, bj, bk;
        brick_coords(brick, (uint32_t)bx, (uint32_t)by, bi, bj, bk);
        uint32_t i = (bi << 2) + (loc & 3u), j = (bj << 2) + ((loc >> 2) & 3u), k = (bk << 2) + (loc >> 4);
        live[v] = slot < voxel_slots && i < (uint32_t)g.ex && j < (uint32_t)g.ey && k < (uint32_t)g.ez;
        qx[v] = isx * (float)i + itx;
        qy[v] = isy * (float)j + ity;
        qz[v] = isz * (float)k + itz;
        best[v] = __builtin_inff();
        besti[v] = 0xFFFFFFFFu;
        besto[v] = 0xFFFFFFFFu;
    }
    for (uint32_t t0 = 0; t0 < n; t0 += kVoxTile) {
        __syncthreads();
        for (int s = threadIdx.x; s < kVoxTile; s += 256) {
            uint32_t p = t0 + s;
            tile[s] = p < n ? pos[p] : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
        }
        __syncthreads();
        const int cnt = min((uint32_t)kVoxTile, n - t0);
        for (int s = 0; s < cnt; ++s) {
            float4 p = tile[s];
#pragma unroll
            for (int v = 0; v < kVoxVPT; ++v) {
                float dx = p.x - qx[v], dy = p.y - qy[v], dz = p.z - qz[v];
                float d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 <= best[v]) { /* NaN (non-finite point) never enters */
                    uint32_t o = __float_as_uint(p.w);
                    if (d2 < best[v] || o < besto[v]) {
                        best[v] = d2;
                        besti[v] = t0 + s;
                        besto[v] = o;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < kVoxVPT; ++v) {
        uint64_t slot = base + (uint64_t)v * 256 + threadIdx.x;
        if (slot < voxel_slots) voxels[slot] = live[v] ? besti[v] : 0u;
        /* brick minimum over the live voxel centres: d2 >= 0, so f32 bit patterns order like values */
        uint32_t b = live[v] ? __float_as_uint(best[v]) : 0x7F7F7F7Fu;
        for (int off = 32; off > 0; off >>= 1) b = min(b, (uint32_t)__shfl_down((int)b, off));
        if ((threadIdx.x & 63) == 0 && slot < voxel_slots) brick_min_bits[slot >> 6] = b;
    }
}

/* Exact NN per voxel by kd-tree descent with bounding-box pruning. A subtree is skipped only when the
 * squared distance from the query to its box EXCEEDS the best d2; per axis the box distance is
 * max(0, lo - q, q - hi), and rounding is monotone, so fl(box d2) <= fl(d2) for every point inside: an
 * equal-distance lower index is never lost. Equal d2 goes to the lowest original index. A wave owns one
 * 4x4x4 brick, so its 64 lanes walk nearly the same path. (node, bound) stack in LDS. */
constexpr int kKdStack = 24;

__device__ __forceinline__ float box_d2(const float q[3], float4 lo, float4 hi) {
    float dx = fmaxf(fmaxf(lo.x - q[0], q[0] - hi.x), 0.f);
    float dy = fmaxf(fmaxf(lo.y - q[1], q[1] - hi.y), 0.f);
    float dz = fmaxf(fmaxf(lo.z - q[2], q[2] - hi.z), 0.f);
    return (dx * dx + dy * dy) + dz * dz;
}

__global__ __launch_bounds__(256) void k_voxel_nn_kd(const float4* __restrict__ kd_lo,
                                                     const float4* __restrict__ kd_hi,
                                                     const float4* __restrict__ pts,
                                                     const uint32_t* __restrict__ new2orig, grid_t g,
                                                     uint64_t voxel_slots, int bx, int by,
                                                     uint32_t* __restrict__ voxels,
                                                     uint32_t* __restrict__ brick_min_bits) {
    __shared__ uint32_t st_node[kKdStack][256];
    __shared__ float st_bound[kKdStack][256];
    const uint64_t slot = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t brick = (uint32_t)(slot >> 6), loc = (uint32_t)(slot & 63);
    uint32_t bi, bj, bk;
    brick_coords(brick, (uint32_t)bx, (uint32_t)by, bi, bj, bk);
    const uint32_t i = (bi << 2) + (loc & 3u), j = (bj << 2) + ((loc >> 2) & 3u), k = (bk << 2) + (loc >> 4);
    const bool live = slot < voxel_slots && i < (uint32_t)g.ex && j < (uint32_t)g.ey && k < (uint32_t)g.ez;
    const float isx = 1.0f / g.sx, isy = 1.0f / g.sy, isz = 1.0f / g.sz;
    float q[3];
    q[0] = isx * (float)i + (-(g.tx * isx));
    q[1] = isy * (float)j + (-(g.ty * isy));
    q[2] = isz * (float)k + (-(g.tz * isz));
    float best = __builtin_inff();
    uint32_t besti = 0xFFFFFFFFu, besto = 0xFFFFFFFFu;
    if (live) {
        int sp = 1;
        st_node[0][threadIdx.x] = 0u;
        st_bound[0][threadIdx.x] = 0.f;
        while (sp > 0) {
            --sp;
            uint32_t node = st_node[sp][threadIdx.x];
            if (st_bound[sp][threadIdx.x] > best) continue;
            for (;;) {
                const uint32_t a = __float_as_uint(kd_lo[node].w), b = __float_as_uint(kd_hi[node].w);
                if (a & 0x80000000u) { /* leaf: points [begin, end) */
                    for (uint32_t p = a & 0x7FFFFFFFu; p < b; ++p) {
                        float4 P = pts[p];
                        float dx = P.x - q[0], dy = P.y - q[1], dz = P.z - q[2];
                        float d2 = (dx * dx + dy * dy) + dz * dz;
                        if (d2 <= best) {
                            uint32_t ni = __float_as_uint(P.w);
                            uint32_t o = new2orig[ni];
                            if (d2 < best || o < besto) {
                                best = d2;
                                besti = ni;
                                besto = o;
                            }
                        }
                    }
                    break;
                }
                /* internal: a = left child, b = right child; go to the nearer box, keep the other */
                float dl = box_d2(q, kd_lo[a], kd_hi[a]);
                float dr = box_d2(q, kd_lo[b], kd_hi[b]);
                uint32_t nearc = dl <= dr ? a : b, farc = dl <= dr ? b : a;
                float dn = fminf(dl, dr), df = fmaxf(dl, dr);
                if (!(df > best) && sp < kKdStack) {
                    st_node[sp][threadIdx.x] = farc;
                    st_bound[sp][threadIdx.x] = df;
                    ++sp;
                }
                if (dn > best) break;
                node = nearc;
            }
        }
    }
    if (slot < voxel_slots) voxels[slot] = live ? besti : 0u;
    uint32_t bmin = live ? __float_as_uint(best) : 0x7F7F7F7Fu;
    for (int off = 32; off > 0; off >>= 1) bmin = min(bmin, (uint32_t)__shfl_down((int)bmin, off));
    if ((threadIdx.x & 63) == 0 && slot < voxel_slots) brick_min_bits[slot >> 6] = bmin;
}

/* The same search with the 64 voxels of a brick walking the tree TOGETHER (a packet): one stack of node ids per wave
 * (256 bytes of LDS instead of 48 KB of per-thread stacks per workgroup), node and leaf loads at one address for the
 * whole wave, and a subtree is entered when ANY lane's box distance does not exceed that lane's best — which is when the
 * per-thread walk would have made the wave execute it anyway, under divergence. A lane that did not ask for a subtree
 * still tests its points: that can only find what its own rule would accept (d2 < best, or d2 == best at a lower
 * original index), so every lane ends with the same (distance, index) as on its own. The order of the two children
 * follows lane 21 (a voxel in the middle of the brick); any order is correct. */
constexpr int kKdPacketStack = 64;
static_assert(kKdPacketStack >= kLbvhMaxDepth + 2, "the packet walk must hold a path of the deepest device-built tree");
__global__ __launch_bounds__(256) void k_voxel_nn_packet(const float4* __restrict__ kd_lo,
                                                         const float4* __restrict__ kd_hi,
                                                         const float4* __restrict__ pts,
                                                         const uint32_t* __restrict__ new2orig, grid_t g,
                                                         uint64_t voxel_slots, int bx, int by,
                                                         uint32_t* __restrict__ voxels,
                                                         uint32_t* __restrict__ brick_min_bits,
                                                         const uint32_t* __restrict__ seed /* per super-brick, or null */,
                                                         const float4* __restrict__ pos /* model points by new index */) {
    __shared__ uint32_t st_node[4][kKdPacketStack];
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t slot = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t brick = (uint32_t)(slot >> 6), loc = (uint32_t)(slot & 63);
    uint32_t bi, bj, bk;
    brick_coords(brick, (uint32_t)bx, (uint32_t)by, bi, bj, bk);
    const uint32_t i = (bi << 2) + (loc & 3u), j = (bj << 2) + ((loc >> 2) & 3u), k = (bk << 2) + (loc >> 4);
    const bool live = slot < voxel_slots && i < (uint32_t)g.ex && j < (uint32_t)g.ey && k < (uint32_t)g.ez;
    const float isx = 1.0f / g.sx, isy = 1.0f / g.sy, isz = 1.0f / g.sz;
    float q[3];
    q[0] = isx * (float)i + (-(g.tx * isx));
    q[1] = isy * (float)j + (-(g.ty * isy));
    q[2] = isz * (float)k + (-(g.tz * isz));
    float best = __builtin_inff();
    uint32_t besti = 0xFFFFFFFFu, besto = 0xFFFFFFFFu;
    if (seed) {
        /* every lane starts from one real candidate — the model point nearest to the centre of the brick's 16^3-voxel
         * super-brick (k_voxel_seed_queries + one k-nearest walk per super-brick, 1/4096 of the voxels) — instead of
         * from nothing: the descent that only FINDS a first bound is gone, what remains is the part that proves it
         * (11.2 -> 8.5 ms per 1.7e8 voxels; the voxel's own answer as the seed: 8.0). A candidate is a candidate: it
         * loses to any point with a smaller d2, or an equal d2 and a lower original index, and no subtree that could
         * hold such a point is skipped (its box distance does not exceed this bound). */
        const uint32_t si = seed[brick >> 6];
        const float4 P = pos[si];
        const float dx = P.x - q[0], dy = P.y - q[1], dz = P.z - q[2];
        best = (dx * dx + dy * dy) + dz * dz;
        besti = si;
        besto = new2orig[si];
    }
    if (__ballot(live)) { /* wave-uniform: a brick with no voxel in the grid has nothing to find */
        int sp = 1;
        if ((threadIdx.x & 63u) == 0u) st_node[wave][0] = 0u;
        while (sp > 0) {
            --sp;
            uint32_t node = (uint32_t)__builtin_amdgcn_readfirstlane((int)st_node[wave][sp]);
            /* does any lane still need this subtree? (bests have shrunk since it was pushed) */
            if (!__ballot(live && !(box_d2(q, kd_lo[node], kd_hi[node]) > best))) continue;
            for (;;) {
                const uint32_t a = __float_as_uint(kd_lo[node].w), b = __float_as_uint(kd_hi[node].w);
                if (a & 0x80000000u) { /* leaf: points [begin, end), the same ones for every lane */
                    for (uint32_t p = a & 0x7FFFFFFFu; p < b; ++p) {
                        const float4 P = pts[p];
                        const float dx = P.x - q[0], dy = P.y - q[1], dz = P.z - q[2];
                        const float d2 = (dx * dx + dy * dy) + dz * dz;
                        if (d2 <= best) {
                            const uint32_t ni = __float_as_uint(P.w);
                            const uint32_t o = new2orig[ni];
                            if (d2 < best || o < besto) {
                                best = d2;
                                besti = ni;
                                besto = o;
                            }
                        }
                    }
                    break;
                }
                const float dl = box_d2(q, kd_lo[a], kd_hi[a]);
                const float dr = box_d2(q, kd_lo[b], kd_hi[b]);
                const bool any_l = __ballot(live && !(dl > best)) != 0ull;
                const bool any_r = __ballot(live && !(dr > best)) != 0ull;
                if (!any_l && !any_r) break;
                /* nearer child first as the brick's middle voxel sees it */
                const float ml = __shfl(dl, 21), mr = __shfl(dr, 21);
                const bool left_first = ml <= mr;
                uint32_t first = left_first ? a : b, second = left_first ? b : a;
                const bool any_first = left_first ? any_l : any_r, any_second = left_first ? any_r : any_l;
                if (any_first && any_second) {
                    if (sp < kKdPacketStack) {
                        if ((threadIdx.x & 63u) == 0u) st_node[wave][sp] = second;
                        ++sp;
                    }
                    node = first;
                } else {
                    node = any_first ? first : second;
                }
                node = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
            }
        }
    }
    if (slot < voxel_slots) voxels[slot] = live ? besti : 0u;
    uint32_t bmin = live ? __float_as_uint(best) : 0x7F7F7F7Fu;
    for (int off = 32; off > 0; off >>= 1) bmin = min(bmin, (uint32_t)__shfl_down((int)bmin, off));
    if ((threadIdx.x & 63) == 0 && slot < voxel_slots) brick_min_bits[slot >> 6] = bmin;
}

/* the centre voxel (8, 8, 8) of every super-brick as a query point, in model space (the voxel kernels' formula) */
__global__ __launch_bounds__(256) void k_voxel_seed_queries(grid_t g, int bx, int by, uint32_t n_super, float4* __restrict__ q) {
    const uint32_t sb = blockIdx.x * 256 + threadIdx.x;
    if (sb >= n_super) return;
    const uint32_t sk = sb / (uint32_t)(bx * by), sr = sb - sk * (uint32_t)(bx * by);
    const uint32_t sj = sr / (uint32_t)bx, si = sr - sj * (uint32_t)bx;
    const uint32_t i = (si << 4) + 8u, j = (sj << 4) + 8u, k = (sk << 4) + 8u;
    const float isx = 1.0f / g.sx, isy = 1.0f / g.sy, isz = 1.0f / g.sz;
    q[sb] = make_float4(isx * (float)i + (-(g.tx * isx)), isy * (float)j + (-(g.ty * isy)), isz * (float)k + (-(g.tz * isz)), 0.f);
}
/* the nearest point's index (the tree's w) per query */
struct knn_sink_first {
    uint32_t* out;
    __device__ void operator()(uint32_t qi, float4, uint32_t l, const float*, const uint32_t ix[4], float) const {
        if (l == 0u) out[qi] = ix[0];
    }
};

/* second-smallest squared distance from each point to the cloud (the point itself included, as
 * nearestKSearch(pnt, 2) does): same box-pruned kd descent, bound = current second best. STACK > 32 (the device-built
 * tree, deeper than a median kd-tree): node ids only on the stack, the box distance is taken again at the pop.
 * BYW: the queries are the tree's own points in its order (a wave walks neighbouring paths) and row = the original
 * index in their w; otherwise row = query number. */
template <int STACK = kKdStack, bool BYW = false>
__global__ __launch_bounds__(256) void k_knn2(const float4* __restrict__ kd_lo, const float4* __restrict__ kd_hi,
                                              const float4* __restrict__ pts, const float4* __restrict__ queries,
                                              uint32_t n, float* __restrict__ out) {
    constexpr bool kBounds = STACK <= 32;
    __shared__ uint32_t st_node[STACK][256];
    __shared__ float st_bound[kBounds ? STACK : 1][256];
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float4 Q = queries[t];
    float q[3] = {Q.x, Q.y, Q.z};
    float b1 = __builtin_inff(), b2 = __builtin_inff();
    if (Q.x == Q.x) {
        int sp = 1;
        st_node[0][threadIdx.x] = 0u;
        if (kBounds) st_bound[0][threadIdx.x] = 0.f;
        while (sp > 0) {
            --sp;
            uint32_t node = st_node[sp][threadIdx.x];
            if (kBounds ? st_bound[sp][threadIdx.x] > b2 : box_d2(q, kd_lo[node], kd_hi[node]) > b2) continue;
            for (;;) {
                const uint32_t a = __float_as_uint(kd_lo[node].w), b = __float_as_uint(kd_hi[node].w);
                if (a & 0x80000000u) {
                    for (uint32_t p = a & 0x7FFFFFFFu; p < b; ++p) {
                        float4 P = pts[p];
                        float dx = P.x - q[0], dy = P.y - q[1], dz = P.z - q[2];
                        float d2 = (dx * dx + dy * dy) + dz * dz;
                        if (d2 < b1) {
                            b2 = b1;
                            b1 = d2;
                        } else if (d2 < b2) {
                            b2 = d2;
                        }
                    }
                    break;
                }
                float dl = box_d2(q, kd_lo[a], kd_hi[a]);
                float dr = box_d2(q, kd_lo[b], kd_hi[b]);
                uint32_t nearc = dl <= dr ? a : b, farc = dl <= dr ? b : a;
                float dn = fminf(dl, dr), df = fmaxf(dl, dr);
                if (!(df > b2) && sp < STACK) {
                    st_node[sp][threadIdx.x] = farc;
                    if (kBounds) st_bound[sp][threadIdx.x] = df;
                    ++sp;
                }
                if (dn > b2) break;
                node = nearc;
            }
        }
    }
    const float r = (Q.x == Q.x) ? sqrtf(b2) : 0.f; /* non-finite points are not part of the tree */
    if (BYW) {
        if (Q.x == Q.x) out[__float_as_uint(Q.w)] = r;
    } else {
        out[t] = r;
    }
}

/* k nearest neighbours (self included) + principal curvatures, one thread per point. The neighbour list
 * lives in LDS, kept sorted by (d2, original index); a subtree is skipped only when its box distance
 * exceeds the current k-th distance, so equal-distance candidates are always seen. */
constexpr int kCurvK = 32;      /* capacity of the neighbour list (k <= 32) */
constexpr int kCurvThreads = 128;

/* principal curvatures from a finished neighbour list, idx_at(j) = original index of the j-th nearest (reference
 * include/impl/pointcloud.hpp:5-43) */
template <class IdxAt>
__device__ __forceinline__ void curvature_of_list(const float4* __restrict__ nrm, uint32_t self, uint32_t have,
                                                  IdxAt idx_at, float* __restrict__ out) {
    const float4 N0 = nrm[self];
    const float n0[3] = {N0.x, N0.y, N0.z};
    float M[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[r][c] = (r == c ? 1.f : 0.f) - n0[r] * n0[c];
    float cen[3] = {0.f, 0.f, 0.f};
    for (uint32_t j = 0; j < have; ++j) {
        const float4 Nj = nrm[idx_at(j)];
        for (int r = 0; r < 3; ++r) {
            float pr = (M[r][0] * Nj.x + M[r][1] * Nj.y) + M[r][2] * Nj.z;
            cen[r] = cen[r] + (pr - cen[r]) / (float)(j + 1);
        }
    }
    float cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; /* 00 01 02 11 12 22 */
    for (uint32_t j = 0; j < have; ++j) {
        const float4 Nj = nrm[idx_at(j)];
        float d[3];
        for (int r = 0; r < 3; ++r) d[r] = ((M[r][0] * Nj.x + M[r][1] * Nj.y) + M[r][2] * Nj.z) - cen[r];
        cov[0] += d[0] * d[0];
        cov[1] += d[0] * d[1];
        cov[2] += d[0] * d[2];
        cov[3] += d[1] * d[1];
        cov[4] += d[1] * d[2];
        cov[5] += d[2] * d[2];
    }
    float ev[3], vmax[3];
    jacobi_eigen3(cov, ev, vmax);
    const float area_inv = 1.0f / (float)have;
    out[0] = vmax[0];
    out[1] = vmax[1];
    out[2] = vmax[2];
    out[3] = ev[1] * area_inv;
    out[4] = ev[2] * area_inv;
}

template <int STACK = kKdStack>
__global__ __launch_bounds__(kCurvThreads) void k_curvature(const float4* __restrict__ kd_lo,
                                                            const float4* __restrict__ kd_hi,
                                                            const float4* __restrict__ pts /* xyz, orig idx */,
                                                            const float4* __restrict__ nrm /* orig order */,
                                                            uint32_t n /* finite points = entries of pts */,
                                                            uint32_t k, float* __restrict__ curv5 /* zeroed */,
                                                            const float4* __restrict__ queries = nullptr,
                                                            uint32_t n_queries = 0,
                                                            const uint32_t* __restrict__ row_of = nullptr) {
    __shared__ uint32_t st_node[STACK][kCurvThreads];
    /* a deep stack (the device-built tree, kLbvhStack) keeps the node only — with the bounds it would not fit the 64 KB
     * of static LDS next to the neighbour lists — and takes the box distance again when it pops */
    constexpr bool kBounds = STACK <= 32;
    __shared__ float st_bound[kBounds ? STACK : 1][kCurvThreads];
    __shared__ float nb_d2[kCurvK][kCurvThreads];
    __shared__ uint32_t nb_idx[kCurvK][kCurvThreads];
    const uint32_t t = blockIdx.x * kCurvThreads + threadIdx.x;
    if (t >= (queries ? n_queries : n)) return;
    const uint32_t tx = threadIdx.x;
    /* the queries run in kd-tree order: the threads of a wave ask for neighbouring points, walk nearly the same
     * nodes and read the same leaves (in the cloud's own order — arbitrary — every lane goes its own way: 57 ms
     * against 36 ms per 1M points) */
    /* with an explicit query list (tm_cloud_curvature_subset): query t = (xyz, original index), row t of curv5 */
    const float4 Q = queries ? queries[t] : pts[t];
    if (!(Q.x == Q.x)) return; /* a non-finite query point keeps its zero row */
    const uint32_t self = __float_as_uint(Q.w);
    /* row_of: the queries were put in the tree's order (so that a wave walks neighbouring paths whatever order the
     * caller listed them in); row_of[t] = the row the caller expects query t in */
    float* out = curv5 + (size_t)(queries ? (row_of ? row_of[t] : t) : self) * 5;
    float q[3] = {Q.x, Q.y, Q.z};
    uint32_t have = 0;
    int sp = 1;
    st_node[0][tx] = 0u;
    if (kBounds) st_bound[0][tx] = 0.f;
    while (sp > 0) {
        --sp;
        uint32_t node = st_node[sp][tx];
        if (have == k && (kBounds ? st_bound[sp][tx] : box_d2(q, kd_lo[node], kd_hi[node])) > nb_d2[k - 1][tx]) continue;
        for (;;) {
            const uint32_t a = __float_as_uint(kd_lo[node].w), b = __float_as_uint(kd_hi[node].w);
            if (a & 0x80000000u) {
                for (uint32_t p = a & 0x7FFFFFFFu; p < b; ++p) {
                    float4 P = pts[p];
                    float dx = P.x - q[0], dy = P.y - q[1], dz = P.z - q[2];
                    float d2 = (dx * dx + dy * dy) + dz * dz;
                    uint32_t idx = __float_as_uint(P.w);
                    if (have == k) {
                        float wd = nb_d2[k - 1][tx];
                        uint32_t wi = nb_idx[k - 1][tx];
                        if (!(d2 < wd || (d2 == wd && idx < wi))) continue;
                    }
                    /* insertion into the sorted list */
                    int pos = (int)(have < k ? have : k - 1);
                    while (pos > 0) {
                        float pd = nb_d2[pos - 1][tx];
                        uint32_t pi = nb_idx[pos - 1][tx];
                        if (pd < d2 || (pd == d2 && pi < idx)) break;
                        nb_d2[pos][tx] = pd;
                        nb_idx[pos][tx] = pi;
                        --pos;
                    }
                    nb_d2[pos][tx] = d2;
                    nb_idx[pos][tx] = idx;
                    if (have < k) ++have;
                }
                break;
            }
            float dl = box_d2(q, kd_lo[a], kd_hi[a]);
            float dr = box_d2(q, kd_lo[b], kd_hi[b]);
            uint32_t nearc = dl <= dr ? a : b, farc = dl <= dr ? b : a;
            float dn = fminf(dl, dr), df = fmaxf(dl, dr);
            const float worst = have == k ? nb_d2[k - 1][tx] : __builtin_inff();
            if (!(df > worst) && sp < STACK) {
                st_node[sp][tx] = farc;
                if (kBounds) st_bound[sp][tx] = df;
                ++sp;
            }
            if (dn > worst) break;
            node = nearc;
        }
    }
    curvature_of_list(nrm, self, have, [&](uint32_t j) { return nb_idx[j][tx]; }, out);
}

/* ---- the same neighbour lists with EIGHT lanes per query (default, curv_groups = 1): k_knn_groups, tm_knn.h ----
 * the lists come back as rows of 32 indices (knn_sink_rows); k_curvature_lists, one thread per query, turns them into
 * curvatures — the same lists, so the same rows bit for bit */
constexpr int kListRow = kKnnGroupK; /* entries per query in the list buffer */
static_assert(kListRow == kCurvK, "the group's registers hold the whole neighbour list");

/* curvature rows from the lists, one thread per query (row t of the caller's order: row_of[t], or t; without a query
 * list the row is the point's original index) */
__global__ __launch_bounds__(256) void k_curvature_lists(const float4* __restrict__ nrm, uint32_t k,
                                                         const float4* __restrict__ queries, uint32_t n_queries,
                                                         const uint32_t* __restrict__ lists, bool rows_by_index,
                                                         const uint32_t* __restrict__ row_of, uint32_t row_base,
                                                         float* __restrict__ curv5 /* zeroed */) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_queries) return;
    const float4 Q = queries[t];
    if (!(Q.x == Q.x)) return; /* a non-finite query point keeps its zero row */
    const uint32_t self = __float_as_uint(Q.w);
    float* out = curv5 + (size_t)(rows_by_index ? self : (row_of ? row_of[row_base + t] : row_base + t)) * 5;
    const uint32_t* row = lists + (size_t)t * kListRow;
    curvature_of_list(nrm, self, k, [&](uint32_t j) { return row[j]; }, out);
}

/* dense x-fastest view with ORIGINAL indices (what the reference's voxel_data_ holds) for the ABI */
__global__ void k_unbrick(const uint32_t* __restrict__ bricked, const uint32_t* __restrict__ new2orig,
                          grid_t g, int bx, int by, uint64_t count, uint32_t* __restrict__ dense) {
    uint64_t lin = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= count) return;
    const uint64_t exy = (uint64_t)g.ex * g.ey;
    uint32_t k = (uint32_t)(lin / exy);
    uint32_t r = (uint32_t)(lin - (uint64_t)k * exy);
    uint32_t j = r / (uint32_t)g.ex, i = r - j * (uint32_t)g.ex;
    uint32_t brick = brick_of(i, j, k, (uint32_t)bx, (uint32_t)by);
    uint32_t v = bricked[(uint64_t)brick * 64u + (((k & 3u) << 4) | ((j & 3u) << 2) | (i & 3u))];
    dense[lin] = v == 0xFFFFFFFFu ? v : new2orig[v];
}

/* brick code = floor(lower bound of sqrt(min d2) / q), clamped to 255 */
__global__ void k_brick_codes(const uint32_t* __restrict__ bits, uint32_t n, float q,
                              unsigned char* __restrict__ code) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float d2 = __uint_as_float(bits[t]);
    float d = sqrtf(d2) * 0.99999f;
    float c = floorf(d / q);
    code[t] = (unsigned char)(c >= 255.f ? 255u : (c > 0.f ? (uint32_t)c : 0u));
}

/* ---- brick codes under a projector's intrinsic_distance (tm_model_set_metric) ----
 * The scoring kernels test a query q that falls into voxel v against ONE model point, m_v = the voxel's stored
 * nearest neighbour (reference voxel_query, include/impl/model.hpp:180-192), with the traits' intrinsic_distance
 * (plane2: 2-D distance in (u, v), include/impl/plane2_traits.hpp:112-116; cylinder: the same with wrap-around
 * in u, include/impl/cylinder_traits.hpp:176-189). Both distances are 1-Lipschitz in q for a fixed m (they are
 * norms of 1-Lipschitz maps of q - m), and q is within sqrt(3) cells of the voxel's reference corner c_v, so
 *     d(q, m_v) >= d(c_v, m_v) - sqrt(3) cell:
 * the minimum of d(c_v, m_v) over a brick's voxels, quantised, is the point-level code (k_metric_voxel_min).
 * For whole tiles the field D(x) = min over ALL model points of d(x, m) is used: it is 1-Lipschitz in x and a
 * lower bound of every d(x, m_v(x)); it does not depend on w, so it is computed once per (i, j) column of the
 * grid (k_metric_column_min, exhaustive with LDS-staged model tiles) and reduced per brick column. */
__global__ __launch_bounds__(256) void k_metric_voxel_min(const uint32_t* __restrict__ voxels,
                                                          const float4* __restrict__ pos, grid_t g, uint64_t voxel_slots,
                                                          int bx, int by, int kind, float radius,
                                                          uint32_t* __restrict__ brick_bits) {
    const uint64_t slot = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t brick = (uint32_t)(slot >> 6), loc = (uint32_t)(slot & 63);
    uint32_t bi, bj, bk;
    brick_coords(brick, (uint32_t)bx, (uint32_t)by, bi, bj, bk);
    const uint32_t i = (bi << 2) + (loc & 3u), j = (bj << 2) + ((loc >> 2) & 3u), k = (bk << 2) + (loc >> 4);
    const bool live = slot < voxel_slots && i < (uint32_t)g.ex && j < (uint32_t)g.ey && k < (uint32_t)g.ez;
    uint32_t bits = 0x7F7F7F7Fu; /* a large finite float: voxels outside the grid are never asked */
    if (live) {
        const float isx = 1.0f / g.sx, isy = 1.0f / g.sy, isz = 1.0f / g.sz;
        const f3 c = make3(isx * (float)i + (-(g.tx * isx)), isy * (float)j + (-(g.ty * isy)),
                           isz * (float)k + (-(g.tz * isz)));
        const float4 P = pos[voxels[slot]];
        const float d = traits_distance(kind, radius, c, make3(P.x, P.y, P.z));
        bits = d == d ? __float_as_uint(fmaxf(d, 0.f)) : 0u; /* NaN (a non-finite model point): no bound */
    }
    for (int off = 32; off > 0; off >>= 1) bits = min(bits, (uint32_t)__shfl_down((int)bits, off));
    if ((threadIdx.x & 63) == 0 && slot < voxel_slots) brick_bits[slot >> 6] = bits;
}

constexpr int kMetricTile = 2048;
__global__ __launch_bounds__(256) void k_metric_column_min(const float4* __restrict__ pos, uint32_t n, grid_t g,
                                                           int kind, float radius, uint32_t* __restrict__ col_bits) {
    __shared__ float2 s_uv[kMetricTile];
    const uint32_t col = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n_cols = (uint32_t)g.ex * (uint32_t)g.ey;
    const uint32_t j = col / (uint32_t)g.ex, i = col - j * (uint32_t)g.ex;
    const float isx = 1.0f / g.sx, isy = 1.0f / g.sy;
    const f3 c = make3(isx * (float)i + (-(g.tx * isx)), isy * (float)j + (-(g.ty * isy)), 0.f);
    float best = 3.0e38f;
    for (uint32_t t0 = 0; t0 < n; t0 += kMetricTile) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < (uint32_t)kMetricTile; t += 256) {
            const float4 P = t0 + t < n ? pos[t0 + t] : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
            s_uv[t] = make_float2(P.x, P.y);
        }
        __syncthreads();
        if (col < n_cols) {
            const uint32_t m = min((uint32_t)kMetricTile, n - t0);
            for (uint32_t t = 0; t < m; ++t) {
                const float2 P = s_uv[t];
                const float d = traits_distance(kind, radius, c, make3(P.x, P.y, 0.f));
                best = d < best ? d : best; /* NaN never wins */
            }
        }
    }
    if (col < n_cols) col_bits[col] = __float_as_uint(fmaxf(best, 0.f));
}

/* code of every brick from a per-brick (linear = 0) or per-column (linear = 1: the brick's 4x4 columns) field of
 * DISTANCES (not squared): floor(lower bound / q), clamped to 255 */
__global__ void k_metric_codes(const uint32_t* __restrict__ bits, int per_column, grid_t g, int bx, int by,
                               uint32_t n_bricks, float q, unsigned char* __restrict__ code) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bricks) return;
    float d;
    if (!per_column) {
        d = __uint_as_float(bits[b]);
    } else {
        uint32_t bi, bj, bk;
        brick_coords(b, (uint32_t)bx, (uint32_t)by, bi, bj, bk);
        d = 3.0e38f;
        for (uint32_t dj = 0; dj < 4; ++dj)
            for (uint32_t di = 0; di < 4; ++di) {
                const uint32_t i = (bi << 2) + di, j = (bj << 2) + dj;
                if (i < (uint32_t)g.ex && j < (uint32_t)g.ey) d = fminf(d, __uint_as_float(bits[j * (uint32_t)g.ex + i]));
            }
    }
    const float c = floorf(d * 0.99999f / q);
    code[b] = (unsigned char)(c >= 255.f ? 255u : (c > 0.f ? (uint32_t)c : 0u));
}

/* ------------------------------------------------------------------------------------------------
 * a7 — pair enumeration. Ordered pairs (a,b) of subset positions, a-major (reference :100).
 * Thread = one `a`; the b side streams through LDS in chunks.
 * ---------------------------------------------------------------------------------------------- */
constexpr int kPairRecBuf = 512; /* records a wave stages in LDS before it appends them (4 KB per wave) */
constexpr int kPairChunk = 256; /* second points staged per workgroup: n_t = 4k gives 16 x 16 workgroups (64 with 1024) */

struct pair_params {
    float lower, upper;      /* diameter * {min,max}_diameter_factor (unsquared, reference :101-102) */
    float lo0, hi0;          /* feature bounds dim 0 (pass 2) */
    uint32_t steps;
    float angle_step;
    uint32_t r1, r2;
    uint32_t n_t;
};

__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t o) {
    uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}

/* reference include/impl/model.hpp:104-111 (same in pass 2, :126-132) */
__device__ __forceinline__ bool pair_filter(f3 pi, f3 ti, f3 pj, float lower, float upper) {
    f3 d = pj - pi;
    float dist = len(d);
    d = over(d, dist);
    if (dist < lower || dist > upper) return false;
    if (1.f - fabsf(dot3(d, ti)) < 0.01f) return false;
    return true;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void k_pairs(const float4* __restrict__ pos,
                                               const float4* __restrict__ tan4,
                                               const uint32_t* __restrict__ subset, pair_params pp,
                                               uint32_t* __restrict__ bounds_ord /* 8: min4,max4 */,
                                               unsigned long long* __restrict__ counter,
                                               unsigned long long* __restrict__ sort_keys,
                                               unsigned long long cap) {
    __shared__ float4 sp[kPairChunk];
    __shared__ float4 st[kPairChunk];
    const uint32_t a = blockIdx.x * 256 + threadIdx.x;
    const uint32_t b0 = blockIdx.y * kPairChunk;
    const bool live = a < pp.n_t;
    f3 pi = make3(0, 0, 0), ti = make3(0, 0, 0);
    uint32_t ia = 0xFFFFFFFFu;
    if (live) {
        ia = subset[a];
        float4 P = pos[ia], T = tan4[ia];
        pi = make3(P.x, P.y, P.z);
        ti = make3(T.x, T.y, T.z);
    }
    for (int s = threadIdx.x; s < kPairChunk; s += 256) {
        uint32_t b = b0 + s;
        if (b < pp.n_t) {
            uint32_t ib = subset[b];
            float4 P = pos[ib];
            P.w = __uint_as_float(ib);
            sp[s] = P;
            st[s] = tan4[ib];
        }
    }
    __syncthreads();
    const int cnt = (int)min((uint32_t)kPairChunk, pp.n_t > b0 ? pp.n_t - b0 : 0u);
    float mn[4] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    uint32_t nvalid = 0;
    /* EMIT: the records of a wave are staged in LDS and go out kPairRecBuf - 64 or more at a time behind ONE returning
     * atomic on the record counter (1.4e7 records: one atomic each was 2 of this pass's 3 ms, one per wave and turn
     * still 2.6e5 same-address atomics); their order does not matter, the sort follows */
    __shared__ unsigned long long s_rec[EMIT ? 4 : 1][EMIT ? kPairRecBuf : 1];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t fill = 0; /* uniform over the live lanes of the wave */
    auto flush = [&]() {
        if (!EMIT || fill == 0u) return;
        const unsigned long long act = __ballot(1);
        const int leader = __ffsll((long long)act) - 1;
        unsigned long long base = 0ull;
        if ((int)lane == leader) base = atomicAdd(counter, (unsigned long long)fill);
        base = __shfl(base, leader);
        /* the wave's LIVE lanes share the copy (the last wave of the subset is partly idle: striding by 64 would leave
         * the idle lanes' share unwritten) */
        const uint32_t n_act = (uint32_t)__popcll(act), my = (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
        for (uint32_t r = my; r < fill; r += n_act)
            if (base + r < cap) sort_keys[base + r] = s_rec[wave][r];
        fill = 0u;
    };
    if (live) {
        for (int s = 0; s < cnt; ++s) {
            float4 P = sp[s];
            uint32_t ib = __float_as_uint(P.w);
            f3 pj = make3(P.x, P.y, P.z);
            /* reference :104 compares the point indices */
            const bool ok = ib != ia && pair_filter(pi, ti, pj, pp.lower, pp.upper);
            feat4 f = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                float4 T = st[s];
                f = pair_feature(pi, ti, pj, make3(T.x, T.y, T.z));
            }
            if (!EMIT) {
                if (ok) {
                    mn[0] = fminf(mn[0], f.f0);
                    mx[0] = fmaxf(mx[0], f.f0);
                    mn[1] = fminf(mn[1], f.f1);
                    mx[1] = fmaxf(mx[1], f.f1);
                    mn[2] = fminf(mn[2], f.f2);
                    mx[2] = fmaxf(mx[2], f.f2);
                    mn[3] = fminf(mn[3], f.f3);
                    mx[3] = fmaxf(mx[3], f.f3);
                    ++nvalid;
                }
            } else {
                const bool emit = ok && feature_valid(f, pp.lo0, pp.hi0);
                unsigned long long rec = 0ull;
                if (emit) {
                    key4 k = feature_key(f, pp.lo0, pp.hi0, pp.steps, pp.angle_step);
                    uint32_t packed = (k.k0 * pp.r1 + k.k1) * pp.r2 + k.k2;
                    uint32_t rank = a * pp.n_t + (b0 + (uint32_t)s);
                    rec = ((unsigned long long)packed << 32) | (0xFFFFFFFFu - rank);
                }
                const unsigned long long em = __ballot(emit); /* every live lane of the wave is here: same trip count */
                if (em) {
                    if (emit) s_rec[wave][fill + (uint32_t)__popcll(em & ((1ull << lane) - 1ull))] = rec;
                    fill += (uint32_t)__popcll(em);
                    if (fill > (uint32_t)kPairRecBuf - 64u) flush();
                }
            }
        }
        flush();
    }
    if (!EMIT) {
        /* wave reduce, then one atomic per wave and bound */
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            for (int off = 32; off > 0; off >>= 1) {
                mn[d] = fminf(mn[d], __shfl_down(mn[d], off));
                mx[d] = fmaxf(mx[d], __shfl_down(mx[d], off));
            }
        }
        for (int off = 32; off > 0; off >>= 1) nvalid += __shfl_down(nvalid, off);
        if ((threadIdx.x & 63) == 0 && nvalid) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                atomicMin(&bounds_ord[d], f2ord(mn[d]));
                atomicMax(&bounds_ord[4 + d], f2ord(mx[d]));
            }
            atomicAdd(counter, (unsigned long long)nvalid);
        }
    }
}

__global__ void k_mark_heads(const unsigned long long* __restrict__ keys, uint32_t n,
                             uint32_t* __restrict__ flags) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t hi = (uint32_t)(keys[t] >> 32);
    flags[t] = (t == 0 || hi != (uint32_t)(keys[t - 1] >> 32)) ? 1u : 0u;
}

__global__ void k_fill_segments(const unsigned long long* __restrict__ keys,
                                const uint32_t* __restrict__ flags,
                                const uint32_t* __restrict__ seg_of, uint32_t n, uint32_t n_t,
                                const uint32_t* __restrict__ subset, uint32_t* __restrict__ seg_key,
                                uint32_t* __restrict__ seg_begin, uint2* __restrict__ entries) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    unsigned long long k = keys[t];
    uint32_t rank = 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull);
    uint32_t a = rank / n_t, b = rank - a * n_t;
    entries[t] = make_uint2(subset[a], subset[b]);
    if (flags[t]) {
        uint32_t s = seg_of[t];
        seg_key[s] = (uint32_t)(k >> 32);
        seg_begin[s] = t;
    }
}

__device__ __forceinline__ uint4 unpack_key(uint32_t packed, uint32_t r1, uint32_t r2) {
    uint32_t k2 = packed % r2;
    uint32_t q = packed / r2;
    uint32_t k1 = q % r1;
    uint32_t k0 = q / r1;
    return make_uint4(k0, k1, k2, k0);
}

__global__ void k_ht_insert(const uint32_t* __restrict__ seg_key, uint32_t n_keys, uint32_t r1,
                            uint32_t r2, uint32_t* __restrict__ ht_slot, uint4* __restrict__ ht_key,
                            uint32_t cap_mask) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_keys) return;
    uint4 k = unpack_key(seg_key[s], r1, r2);
    uint32_t h = murmur_key4(k.x, k.y, k.z, k.w) & cap_mask;
    for (;;) { /* table is at most half full: terminates */
        uint32_t prev = atomicCAS(&ht_slot[h], 0u, s + 1u);
        if (prev == 0u) {
            ht_key[h] = k;
            return;
        }
        h = (h + 1u) & cap_mask;
    }
}

__global__ void k_query_one(const uint32_t* __restrict__ ht_slot, const uint4* __restrict__ ht_key,
                            uint32_t cap_mask, const uint32_t* __restrict__ seg_begin, uint4 key,
                            uint32_t* __restrict__ out /* begin, count */) {
    key4 k = {key.x, key.y, key.z, key.w};
    uint32_t s = ht_find(ht_slot, ht_key, cap_mask, k);
    if (s == 0xFFFFFFFFu) {
        out[0] = 0;
        out[1] = 0;
    } else {
        out[0] = seg_begin[s];
        out[1] = seg_begin[s + 1] - seg_begin[s];
    }
}

/* host: median-split kd-tree over the finite points (Morton-ordered arrays, `order` = new indices);
 * every node carries the bounding box of its points */
struct kd_builder {
    const std::vector<float4>* pos;
    std::vector<float4> lo, hi;
    std::vector<uint32_t> order; /* owned by the root builder */
    uint32_t* ord = nullptr;     /* order.data() of the root: the halves of a parallel build permute their own
                                    disjoint sub-ranges of it */
    static float u2f(uint32_t u) {
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
    int build(int b, int e, int depth) {
        ord = order.data();
        return e - b >= 65536 ? build_parallel(b, e, depth, 5) : build_seq(b, e, depth);
    }
    /* node box + median split along the widest axis; returns the split position or -1 for a leaf */
    int split(int b, int e, int depth, float4& nlo, float4& nhi) {
        float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        for (int i = b; i < e; ++i) {
            const float4& p = (*pos)[ord[i]];
            const float c[3] = {p.x, p.y, p.z};
            for (int d = 0; d < 3; ++d) {
                mn[d] = std::min(mn[d], c[d]);
                mx[d] = std::max(mx[d], c[d]);
            }
        }
        nlo = make_float4(mn[0], mn[1], mn[2], u2f((uint32_t)b | 0x80000000u));
        nhi = make_float4(mx[0], mx[1], mx[2], u2f((uint32_t)e));
        if (e - b <= 8 || depth >= kKdStack - 2) return -1;
        int dim = 0;
        if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
        if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        if (!(mx[dim] - mn[dim] > 0.f)) return -1;
        int mid = (b + e) / 2;
        auto coord = [&](uint32_t i) { const float4& p = (*pos)[i]; return dim == 0 ? p.x : (dim == 1 ? p.y : p.z); };
        std::nth_element(ord + b, ord + mid, ord + e, [&](uint32_t a, uint32_t c) { return coord(a) < coord(c); });
        return mid;
    }
    int build_seq(int b, int e, int depth) {
        int id = (int)lo.size();
        float4 nlo, nhi;
        const int mid = split(b, e, depth, nlo, nhi);
        lo.push_back(nlo);
        hi.push_back(nhi);
        if (mid < 0) return id;
        int l = build_seq(b, mid, depth + 1);
        int r = build_seq(mid, e, depth + 1);
        lo[id].w = u2f((uint32_t)l);
        hi[id].w = u2f((uint32_t)r);
        return id;
    }
    /* The same tree with the same (pre-order) numbering as build_seq, the two halves of the top `levels`
     * levels built on separate host threads: their sub-ranges of the order array are disjoint, each half fills
     * its own node vectors (ids relative to the half) and the parent splices them behind itself. A 1M-point
     * scene: ~0.2 s -> ~0.02 s of the curvature / resolution / voxel-grid preparation. */
    int build_parallel(int b, int e, int depth, int levels) {
        if (levels == 0 || e - b < 4096) return build_seq(b, e, depth);
        int id = (int)lo.size();
        float4 nlo, nhi;
        const int mid = split(b, e, depth, nlo, nhi);
        lo.push_back(nlo);
        hi.push_back(nhi);
        if (mid < 0) return id;
        kd_builder left, right;
        left.pos = right.pos = pos;
        left.ord = right.ord = ord;
        std::thread th([&]() { left.build_parallel(b, mid, depth + 1, levels - 1); });
        right.build_parallel(mid, e, depth + 1, levels - 1);
        th.join();
        const int nl = (int)left.lo.size();
        auto splice = [&](const kd_builder& kb, int base) {
            for (size_t k = 0; k < kb.lo.size(); ++k) {
                float4 a = kb.lo[k], c = kb.hi[k];
                uint32_t wa, wc;
                memcpy(&wa, &a.w, 4);
                memcpy(&wc, &c.w, 4);
                if (!(wa & 0x80000000u)) { /* internal node: its child links are relative to the half */
                    a.w = u2f(wa + (uint32_t)base);
                    c.w = u2f(wc + (uint32_t)base);
                }
                lo.push_back(a);
                hi.push_back(c);
            }
        };
        splice(left, id + 1);
        splice(right, id + 1 + nl);
        lo[id].w = u2f((uint32_t)(id + 1));
        hi[id].w = u2f((uint32_t)(id + 1 + nl));
        return id;
    }
};

/* host: bbox / diameter / to_voxel, reference include/impl/model.hpp:24-61 */
static int setup_grid(tm_model* m) {
    const float big = std::numeric_limits<float>::max();
    float mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
    uint32_t kept = 0;
    /* bounding box over the finite points of the subset handed to init() (include/impl/model.hpp:17-38);
     * an empty subset stands for the whole cloud (:17-19) */
    const uint32_t n_sel = m->grid_subset.empty() ? m->n : (uint32_t)m->grid_subset.size();
    for (uint32_t k = 0; k < n_sel; ++k) {
        const uint32_t i = m->grid_subset.empty() ? k : m->grid_subset[k];
        const float* s = m->h_cloud.data() + (size_t)i * 12;
        if (!(fin3(s) && fin3(s + 4) && fin3(s + 9))) continue;
        ++kept;
        for (int d = 0; d < 3; ++d) {
            mn[d] = std::min(mn[d], s[d]);
            mx[d] = std::max(mx[d], s[d]);
        }
    }
    if (!kept) {
        set_error("model has no finite point");
        return TM_ERR_EMPTY;
    }
    f3 range = make3(mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]);
    m->diameter = len(range);
    const float cell = 0.5f * m->resolution;
    const float rg[3] = {range.x, range.y, range.z};
    float ext[3], sc[3];
    for (int d = 0; d < 3; ++d) {
        ext[d] = std::max(rg[d] / cell, 1.f);
        m->extents[d] = (int)(ext[d] + 2.f * (float)TM_VOXEL_MARGIN);
        sc[d] = rg[d] < 1e-5f ? 1.f : ext[d] / rg[d];
        m->bbmin[d] = mn[d];
        m->bbmax[d] = mx[d];
    }
    for (int i = 0; i < 16; ++i) m->to_voxel[i] = 0.f;
    m->to_voxel[0] = sc[0];
    m->to_voxel[5] = sc[1];
    m->to_voxel[10] = sc[2];
    m->to_voxel[15] = 1.f;
    for (int d = 0; d < 3; ++d) {
        float t = sc[d] * (-mn[d]) + (float)TM_VOXEL_MARGIN;
        t -= 0.5f;
        m->to_voxel[12 + d] = t;
    }
    grid_t& g = m->grid;
    g.sx = sc[0];
    g.sy = sc[1];
    g.sz = sc[2];
    g.tx = m->to_voxel[12];
    g.ty = m->to_voxel[13];
    g.tz = m->to_voxel[14];
    g.ex = m->extents[0];
    g.ey = m->extents[1];
    g.ez = m->extents[2];
    g.fex = (float)g.ex;
    g.fey = (float)g.ey;
    g.fez = (float)g.ez;
    m->grid_ready = true;
    return TM_OK;
}

}  // namespace tmx

using namespace tmx;

namespace tmx {
static int model_arrays_on_device(tm_model* m, const float* surfels); /* defined with the tree builder below */
static int cloud_curvature_on_device(tm_ctx* c, const float* surfels, uint32_t n, uint32_t k, const uint32_t* indices,
                                     uint32_t n_idx, float* curv5);
static int cloud_nn_distances_on_device(tm_ctx* c, const float* surfels, uint32_t n, float* dist, uint32_t* n_fin);
}

extern "C" {

int tm_model_create(tm_ctx* ctx, const float* surfels, uint32_t n, float resolution,
                    float distance_step_count, float angle_step, tm_model** out) {
    TM_GUARD(ctx);
    TM_REQUIRE(ctx && surfels && out, TM_ERR_INVALID, "tm_model_create: null argument");
    TM_REQUIRE(n > 0, TM_ERR_INVALID, "tm_model_create: empty cloud");
    TM_REQUIRE(resolution > 0.f, TM_ERR_INVALID, "tm_model_create: resolution must be > 0");
    TM_HIP_CHECK(hipSetDevice(ctx->device));
    tm_model* m = new tm_model();
    m->ctx = ctx;
    m->n = n;
    m->resolution = resolution;
    m->distance_step_count = distance_step_count;
    m->angle_step = angle_step;
    m->h_cloud.assign(surfels, surfels + (size_t)n * 12);
    /* Round 4: the arrays, their curve order and the tree are built on the device from the raw cloud (the scene's
     * kernels: curve keys, radix sort, linear BVH over the sorted keys) — 22 ms of host sorting, median splits and
     * array shuffling per 100k points became 2. The order is the one the host code below produces (same keys, stable
     * sort). The tree is deeper than a median kd-tree, which only the packet walk of the voxel build tolerates, so
     * voxel_packet = 0 (and model_host_build = 1, the cross-check) keep the host build. */
    if (ctx->opt.voxel_packet && !ctx->opt.model_host_build) {
        int rc = model_arrays_on_device(m, surfels);
        if (rc == TM_OK) rc = setup_grid(m);
        if (rc) {
            tm_model_destroy(m);
            return rc;
        }
        for (int d = 0; d < 4; ++d) {
            m->bmin[d] = std::numeric_limits<float>::max();
            m->bmax[d] = std::numeric_limits<float>::lowest();
        }
        *out = m;
        return TM_OK;
    }
    /* internal order = Hilbert curve over the finite bounding box (non-finite points last); neighbours in
     * space are neighbours in the arrays, so the scoring kernel's gathers of model points share lines.
     * TM_MODEL_MORTON=1 selects the plain Morton order (timing comparisons). */
    {
        const bool hilbert = true;
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (uint32_t i = 0; i < n; ++i) {
            const float* s = surfels + (size_t)i * 12;
            if (!fin3(s)) continue;
            for (int d = 0; d < 3; ++d) {
                lo[d] = std::min(lo[d], (double)s[d]);
                hi[d] = std::max(hi[d], (double)s[d]);
            }
        }
        double ext = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        double inv = ext > 0.0 ? 2097151.0 / ext : 0.0;
        std::vector<std::pair<uint64_t, uint32_t>> order(n);
        for (uint32_t i = 0; i < n; ++i) {
            const float* s = surfels + (size_t)i * 12;
            uint64_t code = ~0ull;
            if (fin3(s)) {
                unsigned int X[3] = {(unsigned int)((s[0] - lo[0]) * inv), (unsigned int)((s[1] - lo[1]) * inv),
                                     (unsigned int)((s[2] - lo[2]) * inv)};
                if (hilbert) {
                    hilbert_transpose21(X);
                    code = (spread21(X[0]) << 2) | (spread21(X[1]) << 1) | spread21(X[2]);
                } else {
                    code = spread21(X[0]) | (spread21(X[1]) << 1) | (spread21(X[2]) << 2);
                }
            }
            order[i] = {code, i};
        }
        std::sort(order.begin(), order.end());
        m->h_new2orig.resize(n);
        m->h_orig2new.resize(n);
        for (uint32_t k = 0; k < n; ++k) {
            m->h_new2orig[k] = order[k].second;
            m->h_orig2new[order[k].second] = k;
        }
    }
    std::vector<float4> pos(n), ref(n), tan4(n), vpos(n);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t i = m->h_new2orig[k];
        const float* s = surfels + (size_t)i * 12;
        f3 t = make3(s[9], s[10], s[11]);
        bool is_t = len(t) > 0.7f; /* reference include/impl/scene.hpp:469-470 */
        pos[k] = make_float4(s[0], s[1], s[2], is_t ? 1.f : 0.f);
        if (!fin3(s)) pos[k].x = std::numeric_limits<float>::quiet_NaN();
        ref[k] = is_t ? make_float4(s[9], s[10], s[11], 1.f) : make_float4(s[4], s[5], s[6], 0.f); /* w = is_tangent */
        tan4[k] = make_float4(s[9], s[10], s[11], 0.f);
        float w;
        memcpy(&w, &i, 4);
        vpos[k] = make_float4(pos[k].x, s[1], s[2], w);
    }
    /* kd-tree for the voxel build */
    std::vector<float4> kd_pts;
    kd_builder kb;
    kb.pos = &pos;
    for (uint32_t k = 0; k < n; ++k)
        if (pos[k].x == pos[k].x) kb.order.push_back(k);
    if (!kb.order.empty()) kb.build(0, (int)kb.order.size(), 0);
    kd_pts.resize(kb.order.size());
    for (size_t k = 0; k < kb.order.size(); ++k) {
        const float4& p = pos[kb.order[k]];
        float w;
        uint32_t ni = kb.order[k];
        memcpy(&w, &ni, 4);
        kd_pts[k] = make_float4(p.x, p.y, p.z, w);
    }
    m->kd_n_nodes = (uint32_t)kb.lo.size();
    m->kd_n_pts = (uint32_t)kd_pts.size();
    int rc;
    if ((rc = m->pos.alloc(n)) || (rc = m->ref.alloc(n)) || (rc = m->tan4.alloc(n)) ||
        (rc = m->vpos.alloc(n)) || (rc = m->new2orig.alloc(n)) || (rc = m->pos3.alloc((size_t)n * 3)) || (rc = m->kd_lo.alloc(kb.lo.size())) || (rc = m->kd_hi.alloc(kb.hi.size())) ||
        (rc = m->kd_pts.alloc(kd_pts.size()))) {
        tm_model_destroy(m);
        return rc;
    }
    std::vector<float> pos3((size_t)n * 3);
    for (uint32_t k = 0; k < n; ++k) {
        pos3[3 * (size_t)k] = pos[k].x;
        pos3[3 * (size_t)k + 1] = pos[k].y;
        pos3[3 * (size_t)k + 2] = pos[k].z;
    }
    hipError_t e = hipMemcpyAsync(m->pos.p, pos.data(), (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(m->pos3.p, pos3.data(), (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && !kb.lo.empty())
        e = hipMemcpyAsync(m->kd_lo.p, kb.lo.data(), kb.lo.size() * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && !kb.hi.empty())
        e = hipMemcpyAsync(m->kd_hi.p, kb.hi.data(), kb.hi.size() * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && !kd_pts.empty())
        e = hipMemcpyAsync(m->kd_pts.p, kd_pts.data(), kd_pts.size() * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(m->vpos.p, vpos.data(), (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(m->new2orig.p, m->h_new2orig.data(), (size_t)n * 4, hipMemcpyHostToDevice,
                           ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(m->ref.p, ref.data(), (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(m->tan4.p, tan4.data(), (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        set_error("tm_model_create: upload failed: %s", hipGetErrorString(e));
        tm_model_destroy(m);
        return TM_ERR_HIP;
    }
    rc = setup_grid(m);
    if (rc) {
        tm_model_destroy(m);
        return rc;
    }
    for (int d = 0; d < 4; ++d) {
        m->bmin[d] = std::numeric_limits<float>::max();
        m->bmax[d] = std::numeric_limits<float>::lowest();
    }
    *out = m;
    return TM_OK;
}

int tm_cloud_resolution(tm_ctx* ctx, const float* surfels, uint32_t n, float* resolution, float* nn_dist) {
    TM_GUARD(ctx);
    TM_REQUIRE(ctx && surfels && resolution, TM_ERR_INVALID, "tm_cloud_resolution: null argument");
    *resolution = 0.f;
    if (n < 2) return TM_OK;
    TM_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->opt.model_host_build) {
        /* round 4: curve order, tree and the 2-NN walk on the device (no kd-tree built on host threads: 3.4 ms per 100k
         * points, 38 ms per million); the running mean stays here, in f32 and in point order (include/common:104-115) */
        std::vector<float> dist(n);
        uint32_t n_fin = 0;
        TM_FORWARD(cloud_nn_distances_on_device(ctx, surfels, n, dist.data(), &n_fin));
        if (n_fin < 2) return TM_OK;
        float accum = 0.f;
        uint32_t i = 0;
        for (uint32_t k = 0; k < n; ++k) {
            if (!fin3(surfels + (size_t)k * 12)) continue;
            accum = accum + (dist[k] - accum) / (float)(++i);
        }
        *resolution = accum;
        if (nn_dist) memcpy(nn_dist, dist.data(), (size_t)n * 4);
        return TM_OK;
    }
    std::vector<float4> pos(n);
    const float nanv = std::numeric_limits<float>::quiet_NaN();
    for (uint32_t i = 0; i < n; ++i) {
        const float* s = surfels + (size_t)i * 12;
        pos[i] = fin3(s) ? make_float4(s[0], s[1], s[2], 0.f) : make_float4(nanv, 0.f, 0.f, 0.f);
    }
    kd_builder kb;
    kb.pos = &pos;
    for (uint32_t k = 0; k < n; ++k)
        if (pos[k].x == pos[k].x) kb.order.push_back(k);
    if (kb.order.size() < 2) return TM_OK;
    kb.build(0, (int)kb.order.size(), 0);
    std::vector<float4> kd_pts(kb.order.size());
    for (size_t k = 0; k < kb.order.size(); ++k) kd_pts[k] = pos[kb.order[k]];
    dbuf<float4> d_lo, d_hi, d_pts, d_q;
    dbuf<float> d_out;
    auto cleanup = [&]() { d_lo.release(); d_hi.release(); d_pts.release(); d_q.release(); d_out.release(); };
    int rc;
    if ((rc = d_lo.alloc(kb.lo.size())) || (rc = d_hi.alloc(kb.hi.size())) || (rc = d_pts.alloc(kd_pts.size())) ||
        (rc = d_q.alloc(n)) || (rc = d_out.alloc(n))) {
        cleanup();
        return rc;
    }
    std::vector<float> dist(n);
    hipError_t e = hipMemcpyAsync(d_lo.p, kb.lo.data(), kb.lo.size() * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_hi.p, kb.hi.data(), kb.hi.size() * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pts.p, kd_pts.data(), kd_pts.size() * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_q.p, pos.data(), (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((k_knn2<kKdStack, false>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_lo.p, d_hi.p, d_pts.p, d_q.p, n,
                           d_out.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(dist.data(), d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) {
        set_error("tm_cloud_resolution: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    /* running mean in f32, in point order (include/common:104-115) */
    float accum = 0.f;
    uint32_t i = 0;
    for (uint32_t k = 0; k < n; ++k) {
        if (!(pos[k].x == pos[k].x)) continue; /* non-finite points are not part of the tree */
        accum = accum + (dist[k] - accum) / (float)(++i);
    }
    *resolution = accum;
    if (nn_dist) memcpy(nn_dist, dist.data(), (size_t)n * 4);
    return TM_OK;
}

int tm_cloud_curvature(tm_ctx* ctx, const float* surfels, uint32_t n, uint32_t k, float* curv5) {
    TM_GUARD(ctx);
    return tm_cloud_curvature_subset(ctx, surfels, n, k, nullptr, n, curv5);
}

/* curvature(k, idx) for the listed points only (indices == NULL: all points, row i = point i). The reference computes
 * it for every point (include/impl/scene.hpp:48-49, model.hpp:68-71) but reads it only where |tangent| > 0.7
 * (:50, :96-99): 4 % of the points at the configs[1] sizes. The neighbours still come from the whole cloud. */
int tm_cloud_curvature_subset(tm_ctx* ctx, const float* surfels, uint32_t n, uint32_t k, const uint32_t* indices,
                              uint32_t n_idx, float* curv5) {
    TM_GUARD(ctx);
    TM_REQUIRE(ctx && surfels && curv5, TM_ERR_INVALID, "tm_cloud_curvature: null argument");
    TM_REQUIRE(k >= 1 && k <= (uint32_t)kCurvK, TM_ERR_UNSUPPORTED, "tm_cloud_curvature: k must be in 1..32");
    if (n == 0 || (indices && n_idx == 0)) return TM_OK;
    if (indices)
        for (uint32_t q = 0; q < n_idx; ++q)
            TM_REQUIRE(indices[q] < n, TM_ERR_INVALID, "tm_cloud_curvature_subset: index out of range");
    TM_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->opt.model_host_build) /* round 4: the cloud goes up raw; order, tree and queries are made there */
        return cloud_curvature_on_device(ctx, surfels, n, k, indices, indices ? n_idx : 0u, curv5);
    const bool dbg = ctx->opt.debug_times != 0; /* where the call's time goes, on stderr */
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(ctx->stream);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[curvature] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::vector<float4> pos(n), nrm(n);
    const float nanv = std::numeric_limits<float>::quiet_NaN();
    for (uint32_t i = 0; i < n; ++i) {
        const float* s = surfels + (size_t)i * 12;
        pos[i] = fin3(s) ? make_float4(s[0], s[1], s[2], 0.f) : make_float4(nanv, 0.f, 0.f, 0.f);
        nrm[i] = make_float4(s[4], s[5], s[6], 0.f);
    }
    kd_builder kb;
    kb.pos = &pos;
    for (uint32_t i = 0; i < n; ++i)
        if (pos[i].x == pos[i].x) kb.order.push_back(i);
    const uint32_t n_rows = indices ? n_idx : n;
    memset(curv5, 0, (size_t)n_rows * 20);
    if (kb.order.empty()) return TM_OK;
    lap("fill");
    kb.build(0, (int)kb.order.size(), 0);
    lap("kd build");
    std::vector<float4> kd_pts(kb.order.size());
    for (size_t i = 0; i < kb.order.size(); ++i) {
        const float4& p = pos[kb.order[i]];
        float w;
        uint32_t oi = kb.order[i];
        memcpy(&w, &oi, 4);
        kd_pts[i] = make_float4(p.x, p.y, p.z, w);
    }
    lap("kd points");
    const uint32_t kk = std::min<uint32_t>(k, (uint32_t)kb.order.size());
    dbuf<float4> d_lo, d_hi, d_pts, d_n, d_q;
    dbuf<float> d_out;
    auto cleanup = [&]() { d_lo.release(); d_hi.release(); d_pts.release(); d_n.release(); d_out.release(); d_q.release(); };
    int rc;
    const uint32_t n_kd = (uint32_t)kd_pts.size();
    std::vector<float4> h_q; /* the listed points as queries: (xyz, original index); non-finite ones keep their zero row */
    std::vector<uint32_t> q_row;
    if (indices) {
        for (uint32_t q = 0; q < n_idx; ++q) {
            const float4& p = pos[indices[q]];
            if (!(p.x == p.x)) continue;
            float w;
            memcpy(&w, &indices[q], 4);
            h_q.push_back(make_float4(p.x, p.y, p.z, w));
            q_row.push_back(q);
        }
        if (h_q.empty()) return TM_OK;
    }
    const uint32_t n_out = indices ? (uint32_t)h_q.size() : n;
    if ((rc = d_lo.alloc(kb.lo.size())) || (rc = d_hi.alloc(kb.hi.size())) || (rc = d_pts.alloc(kd_pts.size())) ||
        (rc = d_n.alloc(n)) || (rc = d_out.alloc((size_t)n_out * 5)) || (indices && (rc = d_q.alloc(h_q.size())))) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d_lo.p, kb.lo.data(), kb.lo.size() * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_hi.p, kb.hi.data(), kb.hi.size() * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pts.p, kd_pts.data(), kd_pts.size() * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_n.p, nrm.data(), (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_out.p, 0, (size_t)n_out * 20, ctx->stream); /* non-finite points stay 0 */
    if (e == hipSuccess && indices)
        e = hipMemcpyAsync(d_q.p, h_q.data(), h_q.size() * 16, hipMemcpyHostToDevice, ctx->stream);
    lap("alloc + upload");
    if (e == hipSuccess) {
        const uint32_t n_threads = indices ? n_out : n_kd;
        hipLaunchKernelGGL(k_curvature<kKdStack>, dim3((n_threads + kCurvThreads - 1) / kCurvThreads), dim3(kCurvThreads), 0, ctx->stream,
                           d_lo.p, d_hi.p, d_pts.p, d_n.p, n_kd, kk, d_out.p, indices ? d_q.p : (const float4*)nullptr, n_out);
        e = hipGetLastError();
    }
    lap("kernel");
    std::vector<float> compact;
    if (indices) compact.resize((size_t)n_out * 5);
    if (e == hipSuccess)
        e = hipMemcpyAsync(indices ? compact.data() : curv5, d_out.p, (size_t)n_out * 20, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && indices)
        for (uint32_t q = 0; q < n_out; ++q) memcpy(curv5 + 5 * (size_t)q_row[q], compact.data() + 5 * (size_t)q, 20);
    lap("download");
    cleanup();
    if (e != hipSuccess) {
        set_error("tm_cloud_curvature: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

}  // extern "C" (reopened below)

namespace tmx {

/* A tree in kd_builder's node format built on the device over points that are already sorted by a curve key whose
 * 3-bit prefix steps are octree cells (a scene's Hilbert keys): a node is a range of the sorted points and splits
 * where the highest bit in which its first and last key differ flips (a linear BVH; equal keys: in the middle), so every
 * node is the content of one octree cell or of a run of sibling cells and its bounding box is tight. Halving the
 * ranges by index instead — simpler, perfectly balanced — visits 6.5 x the leaves per 30-neighbour query (measured:
 * 111 against 17 for a widest-axis median kd-tree; this tree: 14), because a range cut at an arbitrary index
 * straddles cells. Level by level: a queue of (node, range) per level, children numbered from one counter (so the
 * levels are consecutive id ranges), leaves of at most 8 points or at depth kLbvhMaxDepth; boxes bottom-up, one
 * launch per level. ~140 tiny launches, no host synchronisation; the host build of a 1M-point kd-tree was 38 ms plus
 * 30 ms of array shuffling and uploads. The depth is not bounded by log2(n): the consumer needs a stack of
 * kLbvhMaxDepth + 2 (k_curvature<kLbvhStack>). */
__global__ void k_lbvh_init(uint32_t n, uint4* __restrict__ q0, uint32_t* __restrict__ counters /* nodes, |q0|, |q1| */,
                            uint32_t* __restrict__ level_start) {
    q0[0] = make_uint4(0u, 0u, n, 0u);
    counters[0] = 1u;
    counters[1] = 1u;
    counters[2] = 0u;
    level_start[0] = 0u;
    level_start[1] = 1u;
}
__global__ __launch_bounds__(256) void k_lbvh_level(const unsigned long long* __restrict__ keys, const uint4* __restrict__ cur,
                                                    const uint32_t* __restrict__ n_cur, uint4* __restrict__ next,
                                                    uint32_t* __restrict__ n_next, uint32_t* __restrict__ n_nodes,
                                                    float4* __restrict__ lo, float4* __restrict__ hi) {
    const uint32_t n_work = *n_cur;
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < n_work; t += gridDim.x * 256) {
        const uint4 w = cur[t]; /* node, begin, end, depth */
        if (w.z - w.y <= 8u || w.w >= (uint32_t)kLbvhMaxDepth) { /* leaf: the box follows bottom-up */
            lo[w.x].w = __uint_as_float(w.y | 0x80000000u);
            hi[w.x].w = __uint_as_float(w.z);
            continue;
        }
        const unsigned long long k0 = keys[w.y], k1 = keys[w.z - 1u];
        uint32_t mid = (w.y + w.z) / 2u;
        if (k0 != k1) { /* first point whose key has the highest differing bit set */
            const int hb = 63 - __clzll((long long)(k0 ^ k1));
            const unsigned long long bound = (k1 >> hb) << hb;
            uint32_t a = w.y, b = w.z; /* keys[a] < bound <= keys[b - 1] */
            while (a < b) {
                const uint32_t m = (a + b) / 2u;
                if (keys[m] < bound) a = m + 1u; else b = m;
            }
            mid = a;
        }
        const uint32_t c = atomicAdd(n_nodes, 2u), slot = atomicAdd(n_next, 2u);
        lo[w.x].w = __uint_as_float(c);
        hi[w.x].w = __uint_as_float(c + 1u);
        next[slot] = make_uint4(c, w.y, mid, w.w + 1u);
        next[slot + 1u] = make_uint4(c + 1u, mid, w.z, w.w + 1u);
    }
}
/* after level d: the nodes numbered so far end level d + 1; the queue just read becomes the next output */
__global__ void k_lbvh_mark(uint32_t* __restrict__ counters, uint32_t consumed, uint32_t* __restrict__ level_end) {
    *level_end = counters[0];
    counters[1 + consumed] = 0u;
}
/* level_start[from + 1 .. to] = level_start[from] (the levels a shortened build never reached are empty) */
__global__ void k_lbvh_fill_starts(uint32_t* __restrict__ level_start, uint32_t from, uint32_t to) {
    const uint32_t v = level_start[from];
    for (uint32_t d = from + 1u + threadIdx.x; d <= to; d += blockDim.x) level_start[d] = v;
}
__global__ __launch_bounds__(256) void k_lbvh_boxes(const float4* __restrict__ pts, const uint32_t* __restrict__ level_start,
                                                    float4* __restrict__ lo, float4* __restrict__ hi) {
    const uint32_t n0 = level_start[0], n1 = level_start[1];
    for (uint32_t id = n0 + blockIdx.x * 256 + threadIdx.x; id < n1; id += gridDim.x * 256) {
        const uint32_t a = __float_as_uint(lo[id].w), b = __float_as_uint(hi[id].w);
        float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        if (a & 0x80000000u) {
            for (uint32_t p = a & 0x7FFFFFFFu; p < b; ++p) {
                const float4 P = pts[p];
                mn[0] = fminf(mn[0], P.x); mx[0] = fmaxf(mx[0], P.x);
                mn[1] = fminf(mn[1], P.y); mx[1] = fmaxf(mx[1], P.y);
                mn[2] = fminf(mn[2], P.z); mx[2] = fmaxf(mx[2], P.z);
            }
        } else {
            const float4 l0 = lo[a], l1 = lo[b], h0 = hi[a], h1 = hi[b];
            mn[0] = fminf(l0.x, l1.x); mn[1] = fminf(l0.y, l1.y); mn[2] = fminf(l0.z, l1.z);
            mx[0] = fmaxf(h0.x, h1.x); mx[1] = fmaxf(h0.y, h1.y); mx[2] = fmaxf(h0.z, h1.z);
        }
        lo[id] = make_float4(mn[0], mn[1], mn[2], __uint_as_float(a));
        hi[id] = make_float4(mx[0], mx[1], mx[2], __uint_as_float(b));
    }
}
/* queues the build of the tree over pts[0, n_fin) (in the order of their sorted keys): lo / hi = 2 * n_fin nodes at
 * most, q0 / q1 = n_fin work items each, cnt = 3 + kLbvhMaxDepth + 3 words (nodes, |q0|, |q1|, level starts; cnt[0]
 * ends up as the node count) */
void lbvh_build(tm_ctx* c, const unsigned long long* d_keys, const float4* d_pts, uint32_t n_fin, float4* d_lo,
                float4* d_hi, uint4* d_q0, uint4* d_q1, uint32_t* d_cnt) {
    uint32_t* level_start = d_cnt + 3;
    hipLaunchKernelGGL(k_lbvh_init, dim3(1), dim3(1), 0, c->stream, n_fin, d_q0, d_cnt, level_start);
    const uint32_t grid = std::min<uint32_t>(1024u, (n_fin + 255u) / 256u);
    /* The tree of a million points is 27-30 levels deep, the cap is 46: the levels are queued in blocks, and after a
     * block the next queue's length comes back (8 bytes, one wait) — an empty queue ends the build instead of ~50 more
     * launches that find nothing to do (3 launches per level, ~5 us each on the host side). The first block covers what
     * a balanced tree of this size needs plus a margin. */
    uint32_t depth = 0; /* levels queued so far */
    uint32_t block = 4u;
    for (uint32_t m = n_fin / 8u; m > 0u; m >>= 1) ++block; /* log2(n / 8) + 4 */
    bool done = false;
    while (!done && depth <= (uint32_t)kLbvhMaxDepth) {
        const uint32_t upto = std::min<uint32_t>(depth + block, (uint32_t)kLbvhMaxDepth + 1u);
        for (uint32_t d = depth; d < upto; ++d) { /* level d's nodes -> level d + 1's */
            const uint32_t in = d & 1u;
            hipLaunchKernelGGL(k_lbvh_level, dim3(grid), dim3(256), 0, c->stream, d_keys, in ? d_q1 : d_q0, d_cnt + 1 + in,
                               in ? d_q0 : d_q1, d_cnt + 1 + (in ^ 1u), d_cnt, d_lo, d_hi);
            hipLaunchKernelGGL(k_lbvh_mark, dim3(1), dim3(1), 0, c->stream, d_cnt, in, level_start + d + 2);
        }
        depth = upto;
        if (depth > (uint32_t)kLbvhMaxDepth) break;
        uint32_t next_len = 1u; /* the queue level `depth` will read */
        if (hipMemcpyAsync(&next_len, d_cnt + 1 + (depth & 1u), 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            next_len = 1u; /* could not look: keep going to the cap, as before */
        done = next_len == 0u;
        block = 6u;
    }
    /* levels beyond `depth` were never numbered: their starts equal the last one (empty ranges for the box passes) */
    if (depth <= (uint32_t)kLbvhMaxDepth)
        hipLaunchKernelGGL(k_lbvh_fill_starts, dim3(1), dim3(64), 0, c->stream, level_start, depth + 1u,
                           (uint32_t)kLbvhMaxDepth + 2u);
    for (uint32_t d = std::min<uint32_t>(depth + 1u, (uint32_t)kLbvhMaxDepth + 2u); d-- > 0;) /* nodes of level d: [level_start[d], level_start[d + 1]) */
        hipLaunchKernelGGL(k_lbvh_boxes, dim3(grid), dim3(256), 0, c->stream, d_pts, level_start + d, d_lo, d_hi);
}

/* the model's arrays in curve order from its raw cloud on the device (what tm_model_create used to assemble on the
 * host): slot k holds original point order[k]; pos.w = ref.w = (|tangent| > 0.7) (reference include/impl/scene.hpp:
 * 469-470), ref = tangent or normal, a non-finite point has pos.x = NaN; kd_pts[k] = (xyz, k) for the finite prefix */
__global__ __launch_bounds__(256) void k_model_gather(const float* __restrict__ cloud, const uint32_t* __restrict__ order,
                                                      uint32_t n, float4* __restrict__ pos, float4* __restrict__ ref,
                                                      float4* __restrict__ tan4, float4* __restrict__ vpos,
                                                      float* __restrict__ pos3, uint32_t* __restrict__ new2orig,
                                                      float4* __restrict__ kd_pts, uint32_t* __restrict__ n_fin) {
    __shared__ uint32_t s_n[4];
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    bool fin = false;
    if (k < n) {
        const uint32_t i = order[k];
        const float4* sp = reinterpret_cast<const float4*>(cloud) + (size_t)i * 3;
        const float4 a = sp[0], b = sp[1], cc = sp[2];
        fin = isfinite(a.x) && isfinite(a.y) && isfinite(a.z);
        const f3 t = make3(cc.y, cc.z, cc.w);
        const bool is_t = len(t) > 0.7f;
        const float px = fin ? a.x : __builtin_nanf("");
        pos[k] = make_float4(px, a.y, a.z, is_t ? 1.f : 0.f);
        ref[k] = is_t ? make_float4(cc.y, cc.z, cc.w, 1.f) : make_float4(b.x, b.y, b.z, 0.f);
        tan4[k] = make_float4(cc.y, cc.z, cc.w, 0.f);
        vpos[k] = make_float4(px, a.y, a.z, __uint_as_float(i));
        pos3[3 * (size_t)k] = px;
        pos3[3 * (size_t)k + 1] = a.y;
        pos3[3 * (size_t)k + 2] = a.z;
        new2orig[k] = i;
        /* non-finite points sort last (k_scene_morton), so the finite ones are a prefix of the order */
        kd_pts[k] = fin ? make_float4(a.x, a.y, a.z, __uint_as_float(k)) : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
    }
    const unsigned long long bl = __ballot(fin);
    if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = (uint32_t)__popcll(bl);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = (s_n[0] + s_n[1]) + (s_n[2] + s_n[3]);
        if (t) atomicAdd(n_fin, t);
    }
}

static int model_arrays_on_device(tm_model* m, const float* surfels) {
    tm_ctx* c = m->ctx;
    const uint32_t n = m->n;
    dbuf<float> d_cloud;
    dbuf<unsigned long long> d_keys;
    dbuf<uint32_t> d_order, d_cnt;
    dbuf<unsigned char> d_bounds;
    dbuf<uint4> d_q;
    auto cleanup = [&]() {
        d_cloud.release();
        d_keys.release();
        d_order.release();
        d_cnt.release();
        d_bounds.release();
        d_q.release();
    };
    int rc;
    if ((rc = d_cloud.alloc((size_t)n * 12)) || (rc = d_bounds.alloc(64)) || (rc = d_cnt.alloc(3 + kLbvhMaxDepth + 3 + 1)) ||
        (rc = d_q.alloc(2 * (size_t)n)) || (rc = m->pos.alloc(n)) || (rc = m->ref.alloc(n)) || (rc = m->tan4.alloc(n)) ||
        (rc = m->vpos.alloc(n)) || (rc = m->new2orig.alloc(n)) || (rc = m->pos3.alloc((size_t)n * 3)) ||
        (rc = m->kd_lo.alloc(2 * (size_t)n)) || (rc = m->kd_hi.alloc(2 * (size_t)n)) || (rc = m->kd_pts.alloc(n))) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d_cloud.p, surfels, (size_t)n * 48, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rc = curve_order(c, d_cloud.p, n, true, d_bounds.p, d_keys, d_order);
        if (rc) {
            cleanup();
            return rc;
        }
    }
    uint32_t* d_nfin = d_cnt.p + 3 + kLbvhMaxDepth + 3;
    uint32_t n_fin = 0, n_nodes = 0;
    if (e == hipSuccess) e = hipMemsetAsync(d_nfin, 0, 4, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_model_gather, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_cloud.p, d_order.p, n, m->pos.p,
                           m->ref.p, m->tan4.p, m->vpos.p, m->pos3.p, m->new2orig.p, m->kd_pts.p, d_nfin);
        e = hipMemcpyAsync(&n_fin, d_nfin, 4, hipMemcpyDeviceToHost, c->stream);
    }
    m->h_new2orig.resize(n);
    if (e == hipSuccess) e = hipMemcpyAsync(m->h_new2orig.data(), d_order.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && n_fin > 0) {
        lbvh_build(c, d_keys.p, m->kd_pts.p, n_fin, m->kd_lo.p, m->kd_hi.p, d_q.p, d_q.p + n, d_cnt.p);
        e = hipMemcpyAsync(&n_nodes, d_cnt.p, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    cleanup();
    if (e != hipSuccess) {
        set_error("tm_model_create: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    m->h_orig2new.resize(n);
    for (uint32_t k = 0; k < n; ++k) m->h_orig2new[m->h_new2orig[k]] = k;
    m->kd_n_nodes = n_nodes;
    m->kd_n_pts = n_fin;
    m->deep_tree = true;
    return TM_OK;
}

/* the finite points of a scene in its curve order (xyz, original index) — mask_ does not matter here — and how many */
__global__ __launch_bounds__(256) void k_kd_scene_points(const float* __restrict__ cloud, const uint32_t* __restrict__ order,
                                                         uint32_t n, float4* __restrict__ pts, uint32_t* __restrict__ n_fin) {
    __shared__ uint32_t s_n[4];
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    bool fin = false;
    if (k < n) {
        const uint32_t i = order[k];
        const float4 a = reinterpret_cast<const float4*>(cloud)[(size_t)i * 3];
        fin = isfinite(a.x) && isfinite(a.y) && isfinite(a.z);
        /* non-finite points sort last (k_scene_morton), so the finite ones are a prefix of the order */
        pts[k] = fin ? make_float4(a.x, a.y, a.z, __uint_as_float(i)) : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
    }
    const unsigned long long b = __ballot(fin);
    if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = (s_n[0] + s_n[1]) + (s_n[2] + s_n[3]);
        if (t) atomicAdd(n_fin, t);
    }
}
/* queues: pts[k] = (xyz, original index) of the k-th point along the curve (NaN x for the non-finite tail), *d_n_fin +=
 * the number of finite points (zero it first) */
void curve_points(tm_ctx* c, const float* d_cloud, const uint32_t* d_order, uint32_t n, float4* d_pts, uint32_t* d_n_fin) {
    hipLaunchKernelGGL(k_kd_scene_points, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_cloud, d_order, n, d_pts, d_n_fin);
}
__global__ __launch_bounds__(256) void k_kd_scene_normals(const float* __restrict__ cloud, uint32_t n, float4* __restrict__ nrm) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 b = reinterpret_cast<const float4*>(cloud)[(size_t)i * 3 + 1];
    nrm[i] = make_float4(b.x, b.y, b.z, 0.f);
}
__global__ __launch_bounds__(256) void k_kd_scene_queries(const float* __restrict__ cloud, const uint32_t* __restrict__ indices,
                                                          uint32_t n_idx, float4* __restrict__ q) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_idx) return;
    const uint32_t i = indices[t];
    const float4 a = reinterpret_cast<const float4*>(cloud)[(size_t)i * 3];
    const bool fin = isfinite(a.x) && isfinite(a.y) && isfinite(a.z);
    q[t] = fin ? make_float4(a.x, a.y, a.z, __uint_as_float(i)) : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
}

/* slot_of[original index] = position in the curve order */
__global__ __launch_bounds__(256) void k_slot_of(const uint32_t* __restrict__ order, uint32_t n, uint32_t* __restrict__ slot_of) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) slot_of[order[k]] = k;
}
/* sort keys of a query list: the curve position of every listed point, value = its place in the list */
__global__ __launch_bounds__(256) void k_query_keys(const uint32_t* __restrict__ idx, uint32_t n_idx,
                                                    const uint32_t* __restrict__ slot_of, uint32_t* __restrict__ keys,
                                                    uint32_t* __restrict__ vals) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_idx) return;
    keys[t] = slot_of[idx[t]];
    vals[t] = t;
}
/* queries in curve order: q[t] = point idx[row_of[t]] */
__global__ __launch_bounds__(256) void k_kd_scene_queries_sorted(const float* __restrict__ cloud, const uint32_t* __restrict__ indices,
                                                                 const uint32_t* __restrict__ row_of, uint32_t n_idx,
                                                                 float4* __restrict__ q) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_idx) return;
    const uint32_t i = indices[row_of[t]];
    const float4 a = reinterpret_cast<const float4*>(cloud)[(size_t)i * 3];
    const bool fin = isfinite(a.x) && isfinite(a.y) && isfinite(a.z);
    q[t] = fin ? make_float4(a.x, a.y, a.z, __uint_as_float(i)) : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
}

/* the work arrays of one curvature call on a scene's device copy; d_out = n_idx rows of 5 floats, valid until the
 * arena is released */
struct scene_curv_work {
    dbuf<unsigned char> arena;
    float* d_out = nullptr;
};

/* kNN-k curvature of the scene points idx[0..n_idx) (host indices, or device indices when idx_on_device) on a linear
 * BVH built over the scene's curve order; rows stay on the device in w.d_out. The caller releases w.arena. */
static int curvature_run(tm_ctx* c, const float* d_cloud, const uint32_t* d_order, const unsigned long long* d_keys,
                         uint32_t n, uint32_t k, const uint32_t* idx, bool idx_on_device, uint32_t n_idx, scene_curv_work& w);
static int scene_curvature_run(tm_scene* s, uint32_t k, const uint32_t* idx, bool idx_on_device, uint32_t n_idx,
                               scene_curv_work& w) {
    return curvature_run(s->ctx, s->d_cloud.p, s->d_order.p, s->d_keys.p, s->n, k, idx, idx_on_device, n_idx, w);
}
/* the same over any raw cloud on the device with its curve order (tmx::curve_order). idx == null: every point, rows
 * by original index (n rows), the queries taken in the tree's order */
static int curvature_run(tm_ctx* c, const float* d_cloud, const uint32_t* d_order, const unsigned long long* d_keys,
                         uint32_t n, uint32_t k, const uint32_t* idx, bool idx_on_device, uint32_t n_idx, scene_curv_work& w) {
    const bool all = idx == nullptr;
    if (all) n_idx = n;
    const bool dbg = c->opt.debug_times != 0; /* where the call's time goes, on stderr */
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(c->stream);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[scene curvature] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    /* one allocation for the call's work arrays (a hipMalloc / hipFree pair per array was 14 ms of the call) */
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    const size_t o_pts = take((size_t)n * 16), o_nrm = take((size_t)n * 16), o_q = take((size_t)n_idx * 16),
                 o_idx = take(((size_t)n_idx + 1) * 4), o_out = take((size_t)n_idx * 20), o_lo = take(2 * (size_t)n * 16),
                 o_hi = take(2 * (size_t)n * 16), o_q0 = take((size_t)n * 16), o_q1 = take((size_t)n * 16),
                 o_cnt = take((3 + kLbvhMaxDepth + 3) * 4);
    /* neighbour lists of one launch's queries (curv_groups): 128 bytes per query, at most kListChunk queries a launch */
    constexpr uint32_t kListChunk = 1u << 20;
    const size_t o_lists = c->opt.curv_groups ? take((size_t)std::min(n_idx, kListChunk) * kListRow * 4) : 0;
    /* subset mode, one thread per query: the queries are walked in the tree's order (a caller's list — ascending
     * original index — is in no spatial order when the cloud is not: every lane of a wave would go its own way). With
     * eight lanes per query a wave walks eight trees whatever their order: the sort (26 launches of a merge sort, 0.2 ms
     * for 36k keys) buys 0.05 ms of walk and is left out */
    const bool reorder = !all && n_idx >= 256u && !c->opt.curv_groups;
    const size_t o_slot = reorder ? take((size_t)n * 4) : 0, o_qk = reorder ? take((size_t)n_idx * 4) : 0,
                 o_qv = reorder ? take((size_t)n_idx * 4) : 0, o_qk2 = reorder ? take((size_t)n_idx * 4) : 0,
                 o_row = reorder ? take((size_t)n_idx * 4) : 0;
    size_t sort_tmp = 0;
    if (reorder) {
        if (rocprim::radix_sort_pairs(nullptr, sort_tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                      (uint32_t*)nullptr, (size_t)n_idx, 0, 32, c->stream) != hipSuccess)
            sort_tmp = 0;
        TM_FORWARD(scratch_ensure(c, sort_tmp));
    }
    dbuf<unsigned char>& arena = w.arena;
    int rc;
    if ((rc = arena.alloc(off))) return rc;
    struct { float4* p; } d_pts{reinterpret_cast<float4*>(arena.p + o_pts)}, d_nrm{reinterpret_cast<float4*>(arena.p + o_nrm)},
        d_q{reinterpret_cast<float4*>(arena.p + o_q)}, d_lo{reinterpret_cast<float4*>(arena.p + o_lo)},
        d_hi{reinterpret_cast<float4*>(arena.p + o_hi)};
    struct { uint32_t* p; } d_idx{reinterpret_cast<uint32_t*>(arena.p + o_idx)}, d_cnt{reinterpret_cast<uint32_t*>(arena.p + o_cnt)};
    struct { float* p; } d_out{reinterpret_cast<float*>(arena.p + o_out)};
    struct { uint4* p; } d_q0{reinterpret_cast<uint4*>(arena.p + o_q0)}, d_q1{reinterpret_cast<uint4*>(arena.p + o_q1)};
    w.d_out = d_out.p;
    uint32_t n_fin = 0;
    hipError_t e = hipMemsetAsync(d_idx.p + n_idx, 0, 4, c->stream);
    if (e == hipSuccess && !all)
        e = hipMemcpyAsync(d_idx.p, idx, (size_t)n_idx * 4, idx_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_out.p, 0, (size_t)n_idx * 20, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_kd_scene_points, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_cloud, d_order, n,
                           d_pts.p, d_idx.p + n_idx);
        hipLaunchKernelGGL(k_kd_scene_normals, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_cloud, n, d_nrm.p);
        if (reorder) {
            uint32_t* d_slot = reinterpret_cast<uint32_t*>(arena.p + o_slot);
            uint32_t* d_qk = reinterpret_cast<uint32_t*>(arena.p + o_qk);
            uint32_t* d_qv = reinterpret_cast<uint32_t*>(arena.p + o_qv);
            uint32_t* d_qk2 = reinterpret_cast<uint32_t*>(arena.p + o_qk2);
            uint32_t* d_row = reinterpret_cast<uint32_t*>(arena.p + o_row);
            hipLaunchKernelGGL(k_slot_of, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_order, n, d_slot);
            hipLaunchKernelGGL(k_query_keys, dim3((n_idx + 255) / 256), dim3(256), 0, c->stream, d_idx.p, n_idx, d_slot, d_qk,
                               d_qv);
            e = rocprim::radix_sort_pairs(c->scratch.p, sort_tmp, d_qk, d_qk2, d_qv, d_row, (size_t)n_idx, 0, 32, c->stream);
            if (e == hipSuccess)
                hipLaunchKernelGGL(k_kd_scene_queries_sorted, dim3((n_idx + 255) / 256), dim3(256), 0, c->stream, d_cloud,
                                   d_idx.p, d_row, n_idx, d_q.p);
        } else if (!all) {
            hipLaunchKernelGGL(k_kd_scene_queries, dim3((n_idx + 255) / 256), dim3(256), 0, c->stream, d_cloud, d_idx.p, n_idx,
                               d_q.p);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&n_fin, d_idx.p + n_idx, 4, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    lap("alloc + points, normals, queries");
    if (e == hipSuccess && n_fin > 0) {
        lbvh_build(c, d_keys, d_pts.p, n_fin, d_lo.p, d_hi.p, d_q0.p, d_q1.p, d_cnt.p);
        lap("tree");
        const uint32_t kk = std::min(k, n_fin);
        const uint32_t n_threads = all ? n_fin : n_idx;
        const uint32_t* d_row_of = reorder ? reinterpret_cast<const uint32_t*>(arena.p + o_row) : (const uint32_t*)nullptr;
        if (c->opt.curv_groups) {
            /* eight lanes per query build the lists, one thread per query turns them into curvatures */
            uint32_t* d_lists = reinterpret_cast<uint32_t*>(arena.p + o_lists);
            const float4* d_queries = all ? d_pts.p : d_q.p; /* all: the tree's own points, every one finite */
            for (uint32_t q0 = 0; q0 < n_threads; q0 += kListChunk) {
                const uint32_t cnt = std::min(kListChunk, n_threads - q0);
                hipLaunchKernelGGL((k_knn_groups<kLbvhStack, knn_queries_f4, knn_sink_rows>), dim3((cnt + 31u) / 32u), dim3(256), 0,
                                   c->stream, d_lo.p, d_hi.p, d_pts.p, kk, knn_queries_f4{d_queries + q0}, cnt,
                                   knn_sink_rows{d_lists});
                hipLaunchKernelGGL(k_curvature_lists, dim3((cnt + 255u) / 256u), dim3(256), 0, c->stream, d_nrm.p, kk,
                                   d_queries + q0, cnt, d_lists, all, d_row_of, q0, d_out.p);
            }
        } else {
            hipLaunchKernelGGL(k_curvature<kLbvhStack>, dim3((n_threads + kCurvThreads - 1) / kCurvThreads), dim3(kCurvThreads),
                               0, c->stream, d_lo.p, d_hi.p, d_pts.p, d_nrm.p, n_fin, kk, d_out.p,
                               all ? (const float4*)nullptr : d_q.p, n_idx, d_row_of);
        }
        e = hipGetLastError();
        lap("kernel");
    }
    if (e != hipSuccess) {
        set_error("scene curvature: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

/* a raw host cloud on the device with its curve order: what the cloud-level calls (curvature, resolution) need before
 * they can use the scene's tree builder */
struct device_cloud {
    dbuf<float> cloud;
    dbuf<unsigned long long> keys;
    dbuf<uint32_t> order;
    dbuf<unsigned char> bounds;
    void release() {
        cloud.release();
        keys.release();
        order.release();
        bounds.release();
    }
};
static int device_cloud_make(tm_ctx* c, const float* surfels, uint32_t n, device_cloud& dc) {
    int rc;
    if ((rc = dc.cloud.alloc((size_t)n * 12)) || (rc = dc.bounds.alloc(64))) return rc;
    TM_HIP_CHECK(hipMemcpyAsync(dc.cloud.p, surfels, (size_t)n * 48, hipMemcpyHostToDevice, c->stream));
    return curve_order(c, dc.cloud.p, n, true, dc.bounds.p, dc.keys, dc.order);
}

static int cloud_curvature_on_device(tm_ctx* c, const float* surfels, uint32_t n, uint32_t k, const uint32_t* indices,
                                     uint32_t n_idx, float* curv5) {
    const uint32_t n_rows = indices ? n_idx : n;
    device_cloud dc;
    scene_curv_work w;
    int rc = device_cloud_make(c, surfels, n, dc);
    if (rc == TM_OK) rc = curvature_run(c, dc.cloud.p, dc.order.p, dc.keys.p, n, k, indices, false, n_idx, w);
    hipError_t e = hipSuccess;
    if (rc == TM_OK) {
        e = hipMemcpyAsync(curv5, w.d_out, (size_t)n_rows * 20, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    w.arena.release();
    dc.release();
    if (rc != TM_OK) return rc;
    if (e != hipSuccess) {
        set_error("tm_cloud_curvature: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

/* distance of every point to its nearest OTHER point of the cloud (the second entry of nearestKSearch(pnt, 2)), rows
 * by original index, 0 for non-finite points; *n_fin = finite points */
static int cloud_nn_distances_on_device(tm_ctx* c, const float* surfels, uint32_t n, float* dist, uint32_t* n_fin_out) {
    device_cloud dc;
    dbuf<unsigned char> arena;
    auto cleanup = [&]() {
        arena.release();
        dc.release();
    };
    int rc = device_cloud_make(c, surfels, n, dc);
    if (rc != TM_OK) {
        cleanup();
        return rc;
    }
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    const size_t o_pts = take((size_t)n * 16), o_lo = take(2 * (size_t)n * 16), o_hi = take(2 * (size_t)n * 16),
                 o_q0 = take((size_t)n * 16), o_q1 = take((size_t)n * 16), o_cnt = take((3 + kLbvhMaxDepth + 3 + 1) * 4),
                 o_out = take((size_t)n * 4);
    if ((rc = arena.alloc(off))) {
        cleanup();
        return rc;
    }
    float4* d_pts = reinterpret_cast<float4*>(arena.p + o_pts);
    float4* d_lo = reinterpret_cast<float4*>(arena.p + o_lo);
    float4* d_hi = reinterpret_cast<float4*>(arena.p + o_hi);
    uint4* d_q0 = reinterpret_cast<uint4*>(arena.p + o_q0);
    uint4* d_q1 = reinterpret_cast<uint4*>(arena.p + o_q1);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(arena.p + o_cnt);
    uint32_t* d_nfin = d_cnt + 3 + kLbvhMaxDepth + 3;
    float* d_out = reinterpret_cast<float*>(arena.p + o_out);
    uint32_t n_fin = 0;
    hipError_t e = hipMemsetAsync(d_nfin, 0, 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, (size_t)n * 4, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_kd_scene_points, dim3((n + 255) / 256), dim3(256), 0, c->stream, dc.cloud.p, dc.order.p, n, d_pts,
                           d_nfin);
        e = hipMemcpyAsync(&n_fin, d_nfin, 4, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && n_fin >= 2) {
        lbvh_build(c, dc.keys.p, d_pts, n_fin, d_lo, d_hi, d_q0, d_q1, d_cnt);
        if (c->opt.curv_groups) /* eight lanes per point, the second entry of a list of two */
            hipLaunchKernelGGL((k_knn_groups<kLbvhStack, knn_queries_f4, knn_sink_kth_by_w>), dim3((n_fin + 31u) / 32u), dim3(256),
                               0, c->stream, d_lo, d_hi, d_pts, 2u, knn_queries_f4{d_pts}, n_fin, knn_sink_kth_by_w{d_out});
        else
            hipLaunchKernelGGL((k_knn2<kLbvhStack, true>), dim3((n_fin + 255) / 256), dim3(256), 0, c->stream, d_lo, d_hi, d_pts,
                               d_pts, n_fin, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(dist, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) {
        set_error("tm_cloud_resolution: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    *n_fin_out = n_fin;
    return TM_OK;
}

/* |tangent| > min_norm [&& ok[i]]: the points whose curvature tangent_mask_ reads (include/impl/scene.hpp:50) */
struct tangent_candidate {
    const float* cloud;
    const unsigned char* ok;
    float min_norm;
    __host__ __device__ bool operator()(uint32_t i) const {
        const float* p = cloud + (size_t)i * 12;
        const float tx = p[9], ty = p[10], tz = p[11];
        const float nrm = sqrtf((tx * tx + ty * ty) + tz * tz); /* Eigen's norm(): sqrt of the left-to-right sum */
        return nrm > min_norm && (!ok || ok[i] != 0);
    }
};

/* mask[cand[q]] = rows == null || !(!(pc_min / pc_max < max_ratio)) — the reference's test `(pc_min / pc_max) < 0.2f`
 * (include/impl/scene.hpp:50-52) with IEEE division; every other entry of the mask was zeroed before */
__global__ __launch_bounds__(256) void k_tangent_mask_from_rows(const uint32_t* __restrict__ cand, uint32_t n_cand,
                                                                const float* __restrict__ rows, float max_ratio,
                                                                int32_t* __restrict__ mask) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= n_cand) return;
    bool t = true;
    if (rows) t = (rows[5 * (size_t)q + 3] / rows[5 * (size_t)q + 4]) < max_ratio;
    mask[cand[q]] = t ? 1 : 0;
}

}  // namespace tmx

extern "C" {

int tm_scene_curvature_subset(tm_scene* s, uint32_t k, const uint32_t* indices, uint32_t n_idx, float* curv5) {
    TM_GUARD(s);
    TM_REQUIRE(s && curv5, TM_ERR_INVALID, "tm_scene_curvature_subset: null argument");
    TM_REQUIRE(k >= 1 && k <= (uint32_t)kCurvK, TM_ERR_UNSUPPORTED, "tm_scene_curvature_subset: k must be in 1..32");
    TM_REQUIRE(indices || n_idx == 0, TM_ERR_INVALID, "tm_scene_curvature_subset: null index list");
    if (n_idx == 0) return TM_OK;
    for (uint32_t q = 0; q < n_idx; ++q)
        TM_REQUIRE(indices[q] < s->n, TM_ERR_INVALID, "tm_scene_curvature_subset: index out of range");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    scene_curv_work w;
    int rc = scene_curvature_run(s, k, indices, false, n_idx, w);
    hipError_t e = hipSuccess;
    if (rc == TM_OK) {
        e = hipMemcpyAsync(curv5, w.d_out, (size_t)n_idx * 20, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    w.arena.release();
    if (rc != TM_OK) return rc;
    if (e != hipSuccess) {
        set_error("tm_scene_curvature_subset: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

/* tangent_mask_ as the reference builds it (include/impl/scene.hpp:46-54), on the scene's device copy:
 *   mask[i] = |tangent(p_i)| > min_norm  [&& ok[i]]  && (k == 0 || pc_min(i) / pc_max(i) < max_ratio)
 * with the kNN-k curvature of exactly the points that pass the first two tests (tm_scene_curvature_subset). The mask
 * becomes the scene's tangent mask (as tm_scene_set_tangent_mask) and comes back to the caller together with the
 * candidates and their curvature rows, so the host keeps nothing to loop over: per million points 1.8 + 3.4 ms of host
 * loops and three calls became one. */
int tm_scene_auto_tangent_mask(tm_scene* s, uint32_t k, float min_norm, float max_ratio, const unsigned char* ok,
                               int32_t* tangent_mask, uint32_t* candidates, float* curv5, uint32_t cap,
                               uint32_t* n_candidates, uint32_t* n_tangent) {
    TM_GUARD(s);
    TM_REQUIRE(s, TM_ERR_INVALID, "tm_scene_auto_tangent_mask: null scene");
    TM_REQUIRE(!s->h_sparse, TM_ERR_UNSUPPORTED, "tm_scene_auto_tangent_mask: not for a projected scene");
    TM_REQUIRE(k <= (uint32_t)kCurvK, TM_ERR_UNSUPPORTED, "tm_scene_auto_tangent_mask: k must be in 0..32");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t n = s->n;
    dbuf<unsigned char> d_ok;
    dbuf<uint32_t> d_cand;
    auto cleanup = [&]() {
        d_ok.release();
        d_cand.release();
    };
    int rc = d_cand.alloc((size_t)n + 1);
    if (rc == TM_OK && ok) rc = d_ok.alloc(n);
    if (rc != TM_OK) {
        cleanup();
        return rc;
    }
    hipError_t e = hipSuccess;
    if (ok) e = hipMemcpyAsync(d_ok.p, ok, n, hipMemcpyHostToDevice, c->stream);
    uint32_t n_cand = 0;
    if (e == hipSuccess) {
        size_t tmp = 0;
        rocprim::counting_iterator<uint32_t> idx(0u);
        auto flags = rocprim::make_transform_iterator(idx, tangent_candidate{s->d_cloud.p, ok ? d_ok.p : nullptr, min_norm});
        e = rocprim::select(nullptr, tmp, idx, flags, d_cand.p, d_cand.p + n, (size_t)n, c->stream);
        if (e == hipSuccess && (rc = scratch_ensure(c, tmp)) != TM_OK) {
            cleanup();
            return rc;
        }
        if (e == hipSuccess) e = rocprim::select(c->scratch.p, tmp, idx, flags, d_cand.p, d_cand.p + n, (size_t)n, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&n_cand, d_cand.p + n, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (e != hipSuccess) {
        cleanup();
        set_error("tm_scene_auto_tangent_mask: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    if (n_candidates) *n_candidates = n_cand;
    if ((candidates || curv5) && cap < n_cand) {
        cleanup();
        set_error("tm_scene_auto_tangent_mask: room for %u candidates, %u found", cap, n_cand);
        return TM_ERR_CAPACITY;
    }
    scene_curv_work w;
    if (k > 0 && n_cand > 0) {
        rc = scene_curvature_run(s, k, d_cand.p, true, n_cand, w);
        if (rc != TM_OK) {
            w.arena.release();
            cleanup();
            return rc;
        }
    }
    e = hipMemsetAsync(s->d_tmask.p, 0, (size_t)n * 4, c->stream);
    if (e == hipSuccess && n_cand > 0) {
        hipLaunchKernelGGL(k_tangent_mask_from_rows, dim3((n_cand + 255) / 256), dim3(256), 0, c->stream, d_cand.p, n_cand,
                           k > 0 ? w.d_out : (const float*)nullptr, max_ratio, s->d_tmask.p);
        e = hipGetLastError();
    }
    s->h_tmask.resize(n);
    if (e == hipSuccess) e = hipMemcpyAsync(s->h_tmask.data(), s->d_tmask.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && candidates && n_cand)
        e = hipMemcpyAsync(candidates, d_cand.p, (size_t)n_cand * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && curv5 && n_cand) {
        if (k > 0) e = hipMemcpyAsync(curv5, w.d_out, (size_t)n_cand * 20, hipMemcpyDeviceToHost, c->stream);
        else memset(curv5, 0, (size_t)n_cand * 20);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    w.arena.release();
    cleanup();
    if (e != hipSuccess) {
        set_error("tm_scene_auto_tangent_mask: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    if (tangent_mask) memcpy(tangent_mask, s->h_tmask.data(), (size_t)n * 4);
    if (n_tangent) {
        uint32_t t = 0;
        for (uint32_t i = 0; i < n; ++i) t += s->h_tmask[i] != 0;
        *n_tangent = t;
    }
    s->tangent_idx_ready = false; /* as tm_scene_set_tangent_mask: the refs of the sorted points follow the mask */
    s->n_sorted = 0;
    return scene_build(s, false);
}

int tm_model_destroy(tm_model* m) {
    if (!m) return TM_OK;
    if (!ctx_alive(m->ctx)) { /* context already gone: the device memory went with it */
        delete m;
        return TM_OK;
    }
    TM_GUARD(m);
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    m->pos.release();
    m->ref.release();
    m->tan4.release();
    m->pos3.release();
    m->vpos.release();
    m->kd_lo.release();
    m->kd_hi.release();
    m->kd_pts.release();
    m->new2orig.release();
    m->voxels.release();
    m->brick_min_bits.release();
    m->brick_code.release();
    m->super_near.release();
    m->brick_code_metric_pt.release();
    m->brick_code_metric_tile.release();
    m->seg_key.release();
    m->seg_begin.release();
    m->entries.release();
    m->ht_slot.release();
    m->ht_key.release();
    delete m;
    return TM_OK;
}

int tm_model_set_grid_subset(tm_model* m, const uint32_t* subset, uint32_t n_subset) {
    TM_GUARD(m);
    TM_REQUIRE(m, TM_ERR_INVALID, "null model");
    TM_REQUIRE(subset || n_subset == 0, TM_ERR_INVALID, "tm_model_set_grid_subset: null subset");
    for (uint32_t k = 0; k < n_subset; ++k)
        TM_REQUIRE(subset[k] < m->n, TM_ERR_INVALID, "tm_model_set_grid_subset: index out of range");
    std::vector<uint32_t> keep = m->grid_subset;
    m->grid_subset.assign(subset, subset + n_subset);
    m->voxels_ready = false; /* the grid changes: the voxel table has to be rebuilt */
    m->h_voxels_cache.clear();
    int rc = setup_grid(m);
    if (rc != TM_OK) {
        m->grid_subset = keep;
        (void)setup_grid(m);
    }
    return rc;
}

int tm_model_build_voxels(tm_model* m) {
    TM_GUARD(m);
    TM_REQUIRE(m, TM_ERR_INVALID, "null model");
    TM_REQUIRE(m->grid_ready, TM_ERR_NOT_INITIALIZED, "model grid not set up");
    tm_ctx* c = m->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    uint64_t count = (uint64_t)m->extents[0] * m->extents[1] * m->extents[2];
    /* the reference's voxel_count is a uint32_t (include/impl/model.hpp:65) */
    TM_REQUIRE(count <= 0xFFFFFFFFull, TM_ERR_UNSUPPORTED, "voxel grid exceeds 2^32 cells");
    m->voxel_count = count;
    for (int d = 0; d < 3; ++d) m->bdim[d] = (m->extents[d] + 15) / 16; /* super-bricks of 16^3 voxels */
    const uint64_t n_bricks64 = (uint64_t)m->bdim[0] * m->bdim[1] * m->bdim[2] * 64ull;
    TM_REQUIRE(n_bricks64 * 64ull <= 0xFFFFFFFFull, TM_ERR_UNSUPPORTED, "voxel grid exceeds 2^32 slots");
    const uint32_t n_bricks = (uint32_t)n_bricks64;
    m->voxel_slots = n_bricks64 * 64ull;
    TM_FORWARD(m->voxels.alloc(m->voxel_slots));
    TM_FORWARD(m->brick_min_bits.alloc(n_bricks));
    TM_FORWARD(m->brick_code.alloc(n_bricks));
    m->super_near_code = 0xFFFFFFFFu; /* the early-drop kernels' super-brick bits are rebuilt on first use */
    m->brick_q = 0.25f * m->resolution;
    uint64_t per_block = 256ull * kVoxVPT;
    uint64_t blocks = (m->voxel_slots + per_block - 1) / per_block;
    TM_REQUIRE(blocks <= 0x7FFFFFFFull, TM_ERR_UNSUPPORTED, "voxel grid too large for one launch");
    /* default: kd-tree descent per voxel; TM_VOXEL_BRUTEFORCE=1 selects the LDS-tiled exhaustive scan
     * (same table, kept as the cross-check and for degenerate clouds without a tree) */
    const bool brute = c->opt.voxel_bruteforce || m->kd_n_nodes == 0;
    if (brute) {
        hipLaunchKernelGGL(k_voxel_nn, dim3((uint32_t)blocks), dim3(256), 0, c->stream, m->vpos.p, m->n,
                           m->grid, m->voxel_slots, m->bdim[0], m->bdim[1], m->voxels.p, m->brick_min_bits.p);
    } else {
        uint64_t kd_blocks = (m->voxel_slots + 255) / 256;
        TM_REQUIRE(kd_blocks <= 0x7FFFFFFFull, TM_ERR_UNSUPPORTED, "voxel grid too large for one launch");
        /* a brick's 64 voxels as one packet (default), or every voxel on its own stack (voxel_packet = 0: the
         * cross-check; identical tables). The model records which tree it was created with: the per-voxel walk's
         * 24-entry stack would silently drop subtrees of the deep device-built tree, so a model that holds one takes
         * the packet walk whatever the option says now (it selects the host tree only at tm_model_create). */
        if (c->opt.voxel_packet || m->deep_tree) {
            /* voxel_seed (default): one candidate per super-brick from a k-nearest walk of its centre */
            dbuf<float4> d_sq;
            dbuf<uint32_t> d_seed;
            const uint32_t n_super = n_bricks >> 6;
            const bool seeded = c->opt.voxel_seed && d_sq.alloc(n_super) == TM_OK && d_seed.alloc(n_super) == TM_OK;
            if (seeded) {
                hipLaunchKernelGGL(k_voxel_seed_queries, dim3((n_super + 255) / 256), dim3(256), 0, c->stream, m->grid, m->bdim[0],
                                   m->bdim[1], n_super, d_sq.p);
                hipLaunchKernelGGL((k_knn_groups<kLbvhStack, knn_queries_f4, knn_sink_first>),
                                   dim3((n_super + (uint32_t)kKnnGroupsPerBlock - 1u) / (uint32_t)kKnnGroupsPerBlock), dim3(256), 0,
                                   c->stream, m->kd_lo.p, m->kd_hi.p, m->kd_pts.p, 1u, knn_queries_f4{d_sq.p}, n_super,
                                   knn_sink_first{d_seed.p});
            }
            hipLaunchKernelGGL(k_voxel_nn_packet, dim3((uint32_t)kd_blocks), dim3(256), 0, c->stream, m->kd_lo.p, m->kd_hi.p,
                               m->kd_pts.p, m->new2orig.p, m->grid, m->voxel_slots, m->bdim[0], m->bdim[1], m->voxels.p,
                               m->brick_min_bits.p, seeded ? d_seed.p : (const uint32_t*)nullptr, m->pos.p);
            d_sq.release(); /* stream-ordered reuse: the blocks go back to the context's cache behind the kernels */
            d_seed.release();
        } else {
            hipLaunchKernelGGL(k_voxel_nn_kd, dim3((uint32_t)kd_blocks), dim3(256), 0, c->stream, m->kd_lo.p, m->kd_hi.p,
                               m->kd_pts.p, m->new2orig.p, m->grid, m->voxel_slots, m->bdim[0], m->bdim[1], m->voxels.p,
                               m->brick_min_bits.p);
        }
    }
    TM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_brick_codes, dim3((n_bricks + 255) / 256), dim3(256), 0, c->stream,
                       m->brick_min_bits.p, n_bricks, m->brick_q, m->brick_code.p);
    TM_HIP_CHECK(hipGetLastError());
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    m->voxels_ready = true;
    m->h_voxels_cache.clear();
    /* a metric declared before the grid existed: its brick codes are due now */
    if (m->metric_kind > TM_TRAITS_PLANE) TM_FORWARD(tm_model_set_metric(m, m->metric_kind, m->metric_radius));
    return TM_OK;
}

int tm_model_build_pairs(tm_model* m, const uint32_t* subset, uint32_t n_t,
                         float min_diameter_factor, float max_diameter_factor) {
    TM_GUARD(m);
    TM_REQUIRE(m && subset, TM_ERR_INVALID, "tm_model_build_pairs: null argument");
    TM_REQUIRE(n_t >= 2, TM_ERR_EMPTY, "tm_model_build_pairs: subset needs at least two points");
    TM_REQUIRE(n_t < 65536, TM_ERR_UNSUPPORTED, "tm_model_build_pairs: subset >= 65536 points");
    for (uint32_t i = 0; i < n_t; ++i)
        TM_REQUIRE(subset[i] < m->n, TM_ERR_INVALID, "tm_model_build_pairs: index out of range");
    tm_ctx* c = m->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    m->pairs_ready = false;
    m->subset.assign(subset, subset + n_t);

    /* key packing radices: k0 < steps; angle bins are bounded by pi / angle_step */
    uint32_t steps = (uint32_t)m->distance_step_count;
    TM_REQUIRE(steps >= 1 && m->angle_step > 0.f, TM_ERR_UNSUPPORTED,
               "distance_step_count < 1 or angle_step <= 0 is not supported");
    double abins = std::floor(3.14159265358979 / (double)m->angle_step) + 2.0;
    TM_REQUIRE(abins < 65536.0 && (double)steps * abins * abins < 4294967296.0, TM_ERR_UNSUPPORTED,
               "key space does not pack into 32 bits");
    pair_params pp;
    pp.lower = m->diameter * min_diameter_factor;
    pp.upper = m->diameter * max_diameter_factor;
    pp.lo0 = pp.hi0 = 0.f;
    pp.steps = steps;
    pp.angle_step = m->angle_step;
    pp.r1 = pp.r2 = (uint32_t)abins;
    pp.n_t = n_t;
    m->r1 = pp.r1;
    m->r2 = pp.r2;

    dbuf<uint32_t> d_subset, d_bounds, d_flags, d_segof;
    dbuf<unsigned long long> d_counter, d_keys_in, d_keys_out;
    int rc = TM_OK;
    auto cleanup = [&]() {
        d_subset.release();
        d_bounds.release();
        d_flags.release();
        d_segof.release();
        d_counter.release();
        d_keys_in.release();
        d_keys_out.release();
    };
#define TM_TRY(expr)      \
    do {                  \
        rc = (expr);      \
        if (rc) {         \
            cleanup();    \
            return rc;    \
        }                 \
    } while (0)
#define TM_TRY_HIP(expr)                                                              \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__,            \
                      hipGetErrorString(_e));                                         \
            cleanup();                                                                \
            return TM_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

    TM_TRY(d_subset.alloc(n_t));
    TM_TRY(d_bounds.alloc(8));
    TM_TRY(d_counter.alloc(1));
    std::vector<uint32_t> subset_new(n_t);
    for (uint32_t i = 0; i < n_t; ++i) subset_new[i] = m->h_orig2new[subset[i]];
    TM_TRY_HIP(hipMemcpyAsync(d_subset.p, subset_new.data(), (size_t)n_t * 4, hipMemcpyHostToDevice, c->stream));
    uint32_t init_bounds[8];
    for (int d = 0; d < 4; ++d) {
        init_bounds[d] = 0xFFFFFFFFu;
        init_bounds[4 + d] = 0u;
    }
    TM_TRY_HIP(hipMemcpyAsync(d_bounds.p, init_bounds, 32, hipMemcpyHostToDevice, c->stream));
    TM_TRY_HIP(hipMemsetAsync(d_counter.p, 0, 8, c->stream));
    dim3 grid((n_t + 255) / 256, (n_t + kPairChunk - 1) / kPairChunk);
    /* pass 1: feature bounds over the filtered pairs (reference :103-118) */
    hipLaunchKernelGGL(k_pairs<false>, grid, dim3(256), 0, c->stream, m->pos.p, m->tan4.p, d_subset.p,
                       pp, d_bounds.p, d_counter.p, (unsigned long long*)nullptr, 0ull);
    TM_TRY_HIP(hipGetLastError());
    uint32_t h_bounds[8];
    unsigned long long h_valid = 0;
    TM_TRY_HIP(hipMemcpyAsync(h_bounds, d_bounds.p, 32, hipMemcpyDeviceToHost, c->stream));
    TM_TRY_HIP(hipMemcpyAsync(&h_valid, d_counter.p, 8, hipMemcpyDeviceToHost, c->stream));
    TM_TRY_HIP(hipStreamSynchronize(c->stream));
    m->n_valid = h_valid;
    if (h_valid == 0) {
        cleanup();
        set_error("no valid model pair: the reference dereferences an empty histogram here");
        return TM_ERR_EMPTY;
    }
    float raw_min[4], raw_max[4];
    for (int d = 0; d < 4; ++d) {
        raw_min[d] = ord2f(h_bounds[d]);
        raw_max[d] = ord2f(h_bounds[4 + d]);
    }
    /* valid_bounds(bounds, pi/10, 2pi, 0, 1): reference :122 */
    tm_valid_bounds(raw_min, raw_max, 0.0f, 1.f, m->bmin, m->bmax);
    pp.lo0 = m->bmin[0];
    pp.hi0 = m->bmax[0];

    /* pass 2: keys -> (packed key, reverse insertion rank) records, then one global radix sort */
    TM_TRY(d_keys_in.alloc(h_valid));
    TM_TRY(d_keys_out.alloc(h_valid));
    TM_TRY_HIP(hipMemsetAsync(d_counter.p, 0, 8, c->stream));
    hipLaunchKernelGGL(k_pairs<true>, grid, dim3(256), 0, c->stream, m->pos.p, m->tan4.p, d_subset.p,
                       pp, d_bounds.p, d_counter.p, d_keys_in.p, (unsigned long long)h_valid);
    TM_TRY_HIP(hipGetLastError());
    unsigned long long h_entries = 0;
    TM_TRY_HIP(hipMemcpyAsync(&h_entries, d_counter.p, 8, hipMemcpyDeviceToHost, c->stream));
    TM_TRY_HIP(hipStreamSynchronize(c->stream));
    if (h_entries > h_valid) h_entries = h_valid; /* cannot happen: pass 2 filters are a subset */
    TM_REQUIRE(h_entries < 0xFFFFFFFFull, TM_ERR_UNSUPPORTED, "pair table exceeds 2^32 entries");
    m->n_entries = h_entries;
    const uint32_t ne = (uint32_t)h_entries;
    if (ne == 0) {
        cleanup();
        set_error("no valid model pair after bounds check");
        return TM_ERR_EMPTY;
    }
    size_t tmp_bytes = 0;
    TM_TRY_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys_in.p, d_keys_out.p, (size_t)ne, 0, 64,
                                        c->stream));
    TM_TRY(scratch_ensure(c, tmp_bytes));
    TM_TRY_HIP(rocprim::radix_sort_keys(c->scratch.p, tmp_bytes, d_keys_in.p, d_keys_out.p, (size_t)ne, 0,
                                        64, c->stream));
    /* segments */
    TM_TRY(d_flags.alloc(ne));
    TM_TRY(d_segof.alloc(ne));
    hipLaunchKernelGGL(k_mark_heads, dim3((ne + 255) / 256), dim3(256), 0, c->stream, d_keys_out.p, ne,
                       d_flags.p);
    TM_TRY_HIP(hipGetLastError());
    size_t scan_bytes = 0;
    TM_TRY_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_flags.p, d_segof.p, 0u, (size_t)ne,
                                       rocprim::plus<uint32_t>(), c->stream));
    TM_TRY(scratch_ensure(c, scan_bytes));
    TM_TRY_HIP(rocprim::exclusive_scan(c->scratch.p, scan_bytes, d_flags.p, d_segof.p, 0u, (size_t)ne,
                                       rocprim::plus<uint32_t>(), c->stream));
    uint32_t last_seg = 0, last_flag = 0;
    TM_TRY_HIP(hipMemcpyAsync(&last_seg, d_segof.p + (ne - 1), 4, hipMemcpyDeviceToHost, c->stream));
    TM_TRY_HIP(hipMemcpyAsync(&last_flag, d_flags.p + (ne - 1), 4, hipMemcpyDeviceToHost, c->stream));
    TM_TRY_HIP(hipStreamSynchronize(c->stream));
    const uint32_t n_keys = last_seg + last_flag;
    m->n_keys = n_keys;
    TM_TRY(m->seg_key.alloc(n_keys));
    TM_TRY(m->seg_begin.alloc((size_t)n_keys + 1));
    TM_TRY(m->entries.alloc(ne));
    hipLaunchKernelGGL(k_fill_segments, dim3((ne + 255) / 256), dim3(256), 0, c->stream, d_keys_out.p,
                       d_flags.p, d_segof.p, ne, n_t, d_subset.p, m->seg_key.p, m->seg_begin.p,
                       m->entries.p);
    TM_TRY_HIP(hipGetLastError());
    TM_TRY_HIP(hipMemcpyAsync(m->seg_begin.p + n_keys, &ne, 4, hipMemcpyHostToDevice, c->stream));
    /* hash index */
    uint32_t cap = 64;
    while (cap < 2u * n_keys) cap <<= 1;
    m->ht_cap = cap;
    TM_TRY(m->ht_slot.alloc(cap));
    TM_TRY(m->ht_key.alloc(cap));
    TM_TRY_HIP(hipMemsetAsync(m->ht_slot.p, 0, (size_t)cap * 4, c->stream));
    hipLaunchKernelGGL(k_ht_insert, dim3((n_keys + 255) / 256), dim3(256), 0, c->stream, m->seg_key.p,
                       n_keys, m->r1, m->r2, m->ht_slot.p, m->ht_key.p, cap - 1);
    TM_TRY_HIP(hipGetLastError());
    m->h_seg_key.resize(n_keys);
    m->h_seg_begin.resize((size_t)n_keys + 1);
    TM_TRY_HIP(hipMemcpyAsync(m->h_seg_key.data(), m->seg_key.p, (size_t)n_keys * 4, hipMemcpyDeviceToHost,
                              c->stream));
    TM_TRY_HIP(hipMemcpyAsync(m->h_seg_begin.data(), m->seg_begin.p, ((size_t)n_keys + 1) * 4,
                              hipMemcpyDeviceToHost, c->stream));
    TM_TRY_HIP(hipStreamSynchronize(c->stream));
    cleanup();
#undef TM_TRY
#undef TM_TRY_HIP
    m->pairs_ready = true;
    return TM_OK;
}

int tm_model_set_metric(tm_model* m, int traits_kind, float radius) {
    TM_GUARD(m);
    TM_REQUIRE(m, TM_ERR_INVALID, "null model");
    TM_REQUIRE(traits_kind >= TM_TRAITS_IDENTITY && traits_kind <= TM_TRAITS_CYLINDER, TM_ERR_INVALID,
               "bad traits kind");
    TM_REQUIRE(traits_kind != TM_TRAITS_CYLINDER || radius > 0.f, TM_ERR_INVALID, "cylinder metric needs a radius");
    m->metric_kind = traits_kind;
    m->metric_radius = radius;
    m->brick_code_metric_pt.release();
    m->brick_code_metric_tile.release();
    if (traits_kind <= TM_TRAITS_PLANE || !m->voxels_ready) return TM_OK; /* 3-D L2: the codes of the voxel build */
    /* brick codes of the near filter and of the tile test under this metric (see k_metric_voxel_min) */
    tm_ctx* c = m->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t n_bricks = (uint32_t)(m->voxel_slots >> 6);
    const uint32_t n_cols = (uint32_t)m->extents[0] * (uint32_t)m->extents[1];
    dbuf<uint32_t> brick_bits, col_bits;
    int rc;
    if ((rc = brick_bits.alloc(n_bricks)) || (rc = col_bits.alloc(n_cols)) ||
        (rc = m->brick_code_metric_pt.alloc(n_bricks)) || (rc = m->brick_code_metric_tile.alloc(n_bricks))) {
        brick_bits.release();
        col_bits.release();
        return rc;
    }
    hipLaunchKernelGGL(k_metric_voxel_min, dim3((uint32_t)((m->voxel_slots + 255) / 256)), dim3(256), 0, c->stream,
                       m->voxels.p, m->pos.p, m->grid, m->voxel_slots, m->bdim[0], m->bdim[1], traits_kind, radius,
                       brick_bits.p);
    hipLaunchKernelGGL(k_metric_column_min, dim3((n_cols + 255) / 256), dim3(256), 0, c->stream, m->pos.p, m->n, m->grid,
                       traits_kind, radius, col_bits.p);
    hipLaunchKernelGGL(k_metric_codes, dim3((n_bricks + 255) / 256), dim3(256), 0, c->stream, brick_bits.p, 0, m->grid,
                       m->bdim[0], m->bdim[1], n_bricks, m->brick_q, m->brick_code_metric_pt.p);
    hipLaunchKernelGGL(k_metric_codes, dim3((n_bricks + 255) / 256), dim3(256), 0, c->stream, col_bits.p, 1, m->grid,
                       m->bdim[0], m->bdim[1], n_bricks, m->brick_q, m->brick_code_metric_tile.p);
    hipError_t e = hipGetLastError();
    /* how many of the grid's bricks carry which code: tells a scoring call whether the filter can remove anything
     * at its threshold (a dense sheet — an unrolled pipe — has no far bricks, and the filter would only cost) */
    std::vector<uint32_t> h_bits(n_bricks);
    std::vector<unsigned char> h_code(n_bricks);
    if (e == hipSuccess)
        e = hipMemcpyAsync(h_bits.data(), brick_bits.p, (size_t)n_bricks * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(h_code.data(), m->brick_code_metric_pt.p, n_bricks, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    brick_bits.release();
    col_bits.release();
    if (e != hipSuccess) {
        set_error("tm_model_set_metric: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    for (uint64_t& v : m->metric_code_hist) v = 0;
    for (uint32_t b = 0; b < n_bricks; ++b)
        if (h_bits[b] != 0x7F7F7F7Fu) ++m->metric_code_hist[h_code[b]]; /* bricks with a voxel inside the grid */
    float cm = 0.f;
    for (int d = 0; d < 3; ++d) cm = std::max(cm, std::max(std::fabs(m->bbmin[d]), std::fabs(m->bbmax[d])));
    m->metric_coord_max = cm;
    return TM_OK;
}

int tm_model_get_info(const tm_model* m, float* diameter, uint32_t* cloud_size, uint32_t* point_count,
                      float* resolution) {
    TM_GUARD(m);
    TM_REQUIRE(m, TM_ERR_INVALID, "null model");
    if (diameter) *diameter = m->diameter;
    if (cloud_size) *cloud_size = m->n;
    if (point_count) *point_count = (uint32_t)m->subset.size();
    if (resolution) *resolution = m->resolution;
    return TM_OK;
}

int tm_model_get_grid(const tm_model* m, int extents[3], float to_voxel16[16]) {
    TM_GUARD(m);
    TM_REQUIRE(m && m->grid_ready, TM_ERR_NOT_INITIALIZED, "model grid not set up");
    if (extents)
        for (int d = 0; d < 3; ++d) extents[d] = m->extents[d];
    if (to_voxel16)
        for (int i = 0; i < 16; ++i) to_voxel16[i] = m->to_voxel[i];
    return TM_OK;
}

int tm_model_download_voxels(const tm_model* m, uint32_t* voxels, uint64_t cap, uint64_t* count) {
    TM_GUARD(m);
    TM_REQUIRE(m, TM_ERR_INVALID, "null model");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    if (count) *count = m->voxel_count;
    if (!voxels) return TM_OK;
    TM_REQUIRE(cap >= m->voxel_count, TM_ERR_CAPACITY, "voxel buffer too small");
    TM_HIP_CHECK(hipSetDevice(m->ctx->device));
    dbuf<uint32_t> dense;
    TM_FORWARD(dense.alloc(m->voxel_count));
    uint64_t blocks = (m->voxel_count + 255) / 256;
    hipLaunchKernelGGL(k_unbrick, dim3((uint32_t)blocks), dim3(256), 0, m->ctx->stream, m->voxels.p,
                       m->new2orig.p, m->grid, m->bdim[0], m->bdim[1], m->voxel_count, dense.p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(voxels, dense.p, m->voxel_count * 4, hipMemcpyDeviceToHost, m->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->ctx->stream);
    dense.release();
    if (e != hipSuccess) {
        set_error("tm_model_download_voxels: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

int tm_model_get_feature_bounds(const tm_model* m, float bmin[4], float bmax[4]) {
    TM_GUARD(m);
    TM_REQUIRE(m && m->pairs_ready, TM_ERR_NOT_INITIALIZED, "pair table not built");
    for (int d = 0; d < 4; ++d) {
        bmin[d] = m->bmin[d];
        bmax[d] = m->bmax[d];
    }
    return TM_OK;
}

int tm_model_get_pair_counts(const tm_model* m, uint64_t* entries, uint64_t* valid_pairs,
                             uint32_t* distinct_keys) {
    TM_GUARD(m);
    TM_REQUIRE(m && m->pairs_ready, TM_ERR_NOT_INITIALIZED, "pair table not built");
    if (entries) *entries = m->n_entries;
    if (valid_pairs) *valid_pairs = m->n_valid;
    if (distinct_keys) *distinct_keys = m->n_keys;
    return TM_OK;
}

int tm_model_download_table(const tm_model* m, uint32_t* keys4, uint32_t* offsets, uint32_t key_cap,
                            uint32_t* pairs_ij, uint64_t pair_cap) {
    TM_GUARD(m);
    TM_REQUIRE(m && m->pairs_ready, TM_ERR_NOT_INITIALIZED, "pair table not built");
    TM_REQUIRE(key_cap >= m->n_keys && pair_cap >= m->n_entries, TM_ERR_CAPACITY,
               "table buffers too small");
    for (uint32_t s = 0; s < m->n_keys; ++s) {
        uint32_t p = m->h_seg_key[s];
        uint32_t k2 = p % m->r2, q = p / m->r2, k1 = q % m->r1, k0 = q / m->r1;
        if (keys4) {
            keys4[4 * s + 0] = k0;
            keys4[4 * s + 1] = k1;
            keys4[4 * s + 2] = k2;
            keys4[4 * s + 3] = k0;
        }
    }
    if (offsets) memcpy(offsets, m->h_seg_begin.data(), ((size_t)m->n_keys + 1) * 4);
    if (pairs_ij) {
        TM_HIP_CHECK(hipSetDevice(m->ctx->device));
        TM_HIP_CHECK(hipMemcpyAsync(pairs_ij, m->entries.p, m->n_entries * 8, hipMemcpyDeviceToHost,
                                    m->ctx->stream));
        TM_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
        for (uint64_t k = 0; k < 2 * m->n_entries; ++k) pairs_ij[k] = m->h_new2orig[pairs_ij[k]];
    }
    return TM_OK;
}

int tm_model_query_key(const tm_model* m, const uint32_t key[4], uint32_t* pairs_ij, uint32_t cap,
                       uint32_t* count_total) {
    TM_GUARD(m);
    TM_REQUIRE(m && key, TM_ERR_INVALID, "tm_model_query: null argument");
    if (!m->pairs_ready) {
        set_error("Cannot query uninitialized model");
        return TM_ERR_NOT_INITIALIZED;
    }
    tm_ctx* c = m->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    uint32_t* d_out = (uint32_t*)c->stats.p; /* 2 words of the ctx counter block */
    hipLaunchKernelGGL(k_query_one, dim3(1), dim3(1), 0, c->stream, m->ht_slot.p, m->ht_key.p,
                       m->ht_cap - 1, m->seg_begin.p, make_uint4(key[0], key[1], key[2], key[3]), d_out);
    TM_HIP_CHECK(hipGetLastError());
    uint32_t h_out[2] = {0, 0};
    TM_HIP_CHECK(hipMemcpyAsync(h_out, d_out, 8, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (count_total) *count_total = h_out[1];
    uint32_t take = std::min(cap, h_out[1]);
    if (pairs_ij && take) {
        TM_HIP_CHECK(hipMemcpyAsync(pairs_ij, m->entries.p + h_out[0], (size_t)take * 8,
                                    hipMemcpyDeviceToHost, c->stream));
        TM_HIP_CHECK(hipStreamSynchronize(c->stream));
        for (uint32_t k = 0; k < 2 * take; ++k) pairs_ij[k] = m->h_new2orig[pairs_ij[k]];
    }
    return TM_OK;
}

int tm_model_query(const tm_model* m, const float feature[4], uint32_t* pairs_ij, uint32_t cap,
                   uint32_t* count_total) {
    TM_GUARD(m);
    TM_REQUIRE(m && feature, TM_ERR_INVALID, "tm_model_query: null argument");
    if (!m->pairs_ready) {
        set_error("Cannot query uninitialized model");
        return TM_ERR_NOT_INITIALIZED;
    }
    uint32_t key[4];
    tm_discretize_feature(feature, m->bmin, m->bmax, m->distance_step_count, m->angle_step, key);
    return tm_model_query_key(m, key, pairs_ij, cap, count_total);
}

int tm_model_voxel_query(const tm_model* m, const float pos4[4], uint32_t* index, int* found) {
    TM_GUARD(m);
    TM_REQUIRE(m && pos4 && index && found, TM_ERR_INVALID, "tm_model_voxel_query: null argument");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    const grid_t& g = m->grid;
    /* to_voxel * pos with the zero entries dropped: s*x + t*w */
    float vx = g.sx * pos4[0] + g.tx * pos4[3];
    float vy = g.sy * pos4[1] + g.ty * pos4[3];
    float vz = g.sz * pos4[2] + g.tz * pos4[3];
    *found = 0;
    if (!(vx > -1.f && vx < g.fex && vy > -1.f && vy < g.fey && vz > -1.f && vz < g.fez)) return TM_OK;
    const uint32_t ix = (uint32_t)(int)vx, iy = (uint32_t)(int)vy, iz = (uint32_t)(int)vz;
    const size_t brick = brick_of((uint32_t)ix, (uint32_t)iy, (uint32_t)iz, (uint32_t)m->bdim[0], (uint32_t)m->bdim[1]);
    size_t lin = brick * 64 + (((iz & 3u) << 4) | ((iy & 3u) << 2) | (ix & 3u));
    uint32_t v = 0;
    TM_HIP_CHECK(hipSetDevice(m->ctx->device));
    TM_HIP_CHECK(hipMemcpyAsync(&v, m->voxels.p + lin, 4, hipMemcpyDeviceToHost, m->ctx->stream));
    TM_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    *index = v < m->n ? m->h_new2orig[v] : v;
    *found = 1;
    return TM_OK;
}

}  // extern "C"
