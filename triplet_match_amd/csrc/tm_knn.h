/* k nearest tree points of a list of queries, EIGHT lanes per query (k <= 32) — the walk behind the curvature pass
 * (reference include/impl/pointcloud.hpp:5-43: nearestKSearch(k) per point), cloud->resolution() (:66-82: the second
 * neighbour's distance) and the neighbour accessors of pointcloud<Point> (include/pointcloud:38-63).
 *
 * One thread per query (k_curvature, k_knn2, k_cloud_knn: kept as cross-checks) leaves a wave executing 64 different
 * tree walks one after the other, each step with one useful lane, and a few ten thousand queries (the tangent
 * candidates of a 1M-point scene) are a few hundred waves: the kernel lasts as long as one wave's 64 walks. Here a
 * query belongs to a group of 8 lanes: a leaf of the device-built tree holds at most 8 points — one per lane —, the
 * sorted list of 32 (d2, index) entries lives in REGISTERS, four consecutive entries per lane, and an insertion is one
 * parallel step (every lane counts its entries below the candidate; the lane where the count ends puts the candidate
 * there and hands its last entry on, the lanes after it shift by one — three 8-lane shuffles, no loop, no LDS). A wave
 * walks 8 trees instead of 64, eight times as many waves share the chip, and a step costs less. The (node, box
 * distance) stack is per group in LDS.
 *
 * The rules are the per-thread kernels': a subtree is skipped only when its box distance EXCEEDS the k-th distance
 * (per axis the box distance is max(0, lo - q, q - hi) and rounding is monotone, so fl(box d2) <= fl(d2) of every point
 * inside: an equal-distance lower index is never lost), candidates are ordered by (d2, index). Unused entries are
 * (inf, 0xFFFFFFFF): "list not full" is "k-th distance = inf", under which nothing is skipped and every point is
 * accepted. The finished list is handed to a sink, each lane with its four entries.
 *
 * Tree lay-out (tmx::lbvh_build, or a host-built median kd-tree): lo[node].xyz / hi[node].xyz = box; inner node:
 * lo.w / hi.w = children; leaf: lo.w = 0x80000000 | first point, hi.w = one past the last; pts = (xyz, index). */
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace tmx {

__device__ __forceinline__ float knn_box_d2(const float q[3], float4 lo, float4 hi) {
    const float dx = fmaxf(fmaxf(lo.x - q[0], q[0] - hi.x), 0.f);
    const float dy = fmaxf(fmaxf(lo.y - q[1], q[1] - hi.y), 0.f);
    const float dz = fmaxf(fmaxf(lo.z - q[2], q[2] - hi.z), 0.f);
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ bool nb_less(float a, uint32_t ai, float b, uint32_t bi) { return a < b || (a == b && ai < bi); }

constexpr int kKnnGroupLanes = 8;  /* lanes per query */
constexpr int kKnnGroupK = 32;     /* list capacity: 8 lanes x 4 registers */
constexpr int kKnnGroupsPerBlock = 32;

/* Queries: float4 operator()(uint32_t qi) -> (xyz, w); x = NaN: no walk (the sink still sees the empty list).
 * Sink: void operator()(uint32_t qi, float4 Q, uint32_t l, const float d[4], const uint32_t ix[4], float kth_d2) — called by
 * all 8 lanes of the group; lane l holds entries 4 l .. 4 l + 3 of the sorted list. */
template <int STACK, class Queries, class Sink>
__global__ __launch_bounds__(256) void k_knn_groups(const float4* __restrict__ kd_lo, const float4* __restrict__ kd_hi,
                                                    const float4* __restrict__ pts, uint32_t k /* 1..32 */, Queries queries,
                                                    uint32_t n_queries, Sink sink) {
    __shared__ uint32_t st_node[STACK][kKnnGroupsPerBlock];
    __shared__ float st_bound[STACK][kKnnGroupsPerBlock];
    const uint32_t l = threadIdx.x & 7u, grp = threadIdx.x >> 3;
    const uint32_t qi = blockIdx.x * (uint32_t)kKnnGroupsPerBlock + grp;
    if (qi >= n_queries) return; /* whole groups leave: the shuffles below stay inside a group */
    const float4 Q = queries(qi);
    const float q[3] = {Q.x, Q.y, Q.z};
    float d[4] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
    uint32_t ix[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    float wd = __builtin_inff(); /* the k-th entry: lane (k - 1) / 4, register (k - 1) % 4 */
    uint32_t wi = 0xFFFFFFFFu;
    const int k_lane = (int)((k - 1u) >> 2), k_reg = (int)((k - 1u) & 3u);
    const uint32_t group_shift = threadIdx.x & 56u; /* the group's first lane within the wave */
    int sp = 0;
    if (Q.x == Q.x) {
        sp = 1;
        if (l == 0u) {
            st_node[0][grp] = 0u;
            st_bound[0][grp] = 0.f;
        }
    }
    while (sp > 0) {
        --sp;
        /* lane 0 of the group wrote the entry, all eight read it: an LDS hand-over inside the wave, fenced so that
         * the compiler keeps the write before the reads (no workgroup barrier: the groups of a wave diverge) */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t node = st_node[sp][grp];
        if (st_bound[sp][grp] > wd) continue;
        for (;;) {
            const uint32_t a = __float_as_uint(kd_lo[node].w), b = __float_as_uint(kd_hi[node].w);
            if (a & 0x80000000u) {
                for (uint32_t p0 = a & 0x7FFFFFFFu; p0 < b; p0 += 8u) { /* a leaf at the depth limit can hold more than 8 */
                    const uint32_t p = p0 + l;
                    const float4 P = pts[p < b ? p : p0];
                    const float dx = P.x - q[0], dy = P.y - q[1], dz = P.z - q[2];
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    const uint32_t idx = __float_as_uint(P.w);
                    const bool cand = p < b && nb_less(d2, idx, wd, wi);
                    uint32_t m = (uint32_t)(__ballot(cand) >> group_shift) & 0xFFu;
                    while (m) {
                        const int j = __ffs((int)m) - 1;
                        m &= m - 1u;
                        const float v = __shfl(d2, j, 8);
                        const uint32_t vi = (uint32_t)__shfl((int)idx, j, 8);
                        if (!nb_less(v, vi, wd, wi)) continue; /* the k-th entry has moved since the ballot */
                        const int c = (int)nb_less(d[0], ix[0], v, vi) + (int)nb_less(d[1], ix[1], v, vi) +
                                      (int)nb_less(d[2], ix[2], v, vi) + (int)nb_less(d[3], ix[3], v, vi);
                        int pc = __shfl_up(c, 1, 8);
                        const float pd = __shfl_up(d[3], 1, 8);
                        const uint32_t pi = (uint32_t)__shfl_up((int)ix[3], 1, 8);
                        if (l == 0u) pc = 4;
                        if (c < 4) {
                            /* pc == 4: the sorted position is in this lane, at c; otherwise c == 0 and the whole lane
                             * shifts, taking the previous lane's last entry */
                            const float ins_d = pc == 4 ? v : pd;
                            const uint32_t ins_i = pc == 4 ? vi : pi;
                            d[3] = c == 3 ? ins_d : d[2];
                            ix[3] = c == 3 ? ins_i : ix[2];
                            d[2] = c > 2 ? d[2] : (c == 2 ? ins_d : d[1]);
                            ix[2] = c > 2 ? ix[2] : (c == 2 ? ins_i : ix[1]);
                            d[1] = c > 1 ? d[1] : (c == 1 ? ins_d : d[0]);
                            ix[1] = c > 1 ? ix[1] : (c == 1 ? ins_i : ix[0]);
                            d[0] = c > 0 ? d[0] : ins_d;
                            ix[0] = c > 0 ? ix[0] : ins_i;
                        }
                        const float kd = k_reg == 0 ? d[0] : (k_reg == 1 ? d[1] : (k_reg == 2 ? d[2] : d[3]));
                        const uint32_t ki = k_reg == 0 ? ix[0] : (k_reg == 1 ? ix[1] : (k_reg == 2 ? ix[2] : ix[3]));
                        wd = __shfl(kd, k_lane, 8);
                        wi = (uint32_t)__shfl((int)ki, k_lane, 8);
                    }
                }
                break;
            }
            const float dl = knn_box_d2(q, kd_lo[a], kd_hi[a]);
            const float dr = knn_box_d2(q, kd_lo[b], kd_hi[b]);
            const uint32_t nearc = dl <= dr ? a : b, farc = dl <= dr ? b : a;
            const float dn = fminf(dl, dr), df = fmaxf(dl, dr);
            if (!(df > wd) && sp < STACK) {
                if (l == 0u) {
                    st_node[sp][grp] = farc;
                    st_bound[sp][grp] = df;
                }
                ++sp;
            }
            if (dn > wd) break;
            node = nearc;
        }
    }
    sink(qi, Q, l, d, ix, wd);
}

/* queries from an array of (xyz, w) */
struct knn_queries_f4 {
    const float4* q;
    __device__ float4 operator()(uint32_t qi) const { return q[qi]; }
};

/* the indices of the list as a row of 32 (the group writes one 128-byte line) */
struct knn_sink_rows {
    uint32_t* rows; /* n_queries x 32 */
    __device__ void operator()(uint32_t qi, float4, uint32_t l, const float*, const uint32_t ix[4], float) const {
        reinterpret_cast<uint4*>(rows)[(size_t)qi * 8u + l] = make_uint4(ix[0], ix[1], ix[2], ix[3]);
    }
};

/* sqrt of the k-th squared distance at row w of the query (cloud->resolution(): k = 2, the queries are the tree's points) */
struct knn_sink_kth_by_w {
    float* out;
    __device__ void operator()(uint32_t, float4 Q, uint32_t l, const float*, const uint32_t*, float kth_d2) const {
        if (l == 0u && Q.x == Q.x) out[__float_as_uint(Q.w)] = sqrtf(kth_d2);
    }
};

}  // namespace tmx
