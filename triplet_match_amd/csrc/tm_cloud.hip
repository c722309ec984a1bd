/*
 * tm_cloud.hip — the neighbour queries of pointcloud<Point> (reference include/pointcloud:38-63,
 * include/impl/pointcloud.hpp:100-210): kdtree() / build_tree_ / set_indices, knn_inclusive / knn_exclusive,
 * radius_search_inclusive / radius_search_exclusive and curvature(float r, idx). The reference answers them
 * with pcl::search::KdTree (FLANN) under a lazily built, mutex-guarded tree; here a tm_cloud handle keeps the
 * points and a box-pruned kd-tree on the device:
 *   k <= 32 neighbours of a BATCH of queries: one thread per query, kd descent with the k-list in LDS
 *     (the engine tm_cloud_curvature uses for its kNN-30);
 *   radius queries and k > 32: one pass over all points (squared distance per point, 4 B written per point),
 *     compaction of the members on the device (rocPRIM select), the few survivors sorted on the host.
 * FLANN's result order is unpinned beyond "ascending distance"; it is DEFINED here as (squared distance, index)
 * ascending, squared distance = (dx*dx + dy*dy) + dz*dz in f32 with d = point - query — the order and the
 * arithmetic of the oracle and of tm_cloud_curvature / tm_scene_radius_subset. Distances handed out are SQUARED
 * (PCL's nearestKSearch / radiusSearch do the same).
 */
#include "tm_internal.h"
#include "tm_knn.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <limits>
#include <mutex>
#include <thread>
#include <utility>

struct tm_cloud {
    tm_ctx* ctx = nullptr;
    uint32_t n = 0;    /* points of the cloud */
    uint32_t n_kd = 0; /* points in the tree: finite, and in the subset when one was given */
    std::vector<float> h_cloud; /* n x 12, for curvature(r, idx) */
    tmx::dbuf<float4> kd_lo, kd_hi; /* node boxes + links (see kd_node_builder) */
    tmx::dbuf<float4> kd_pts;       /* leaf-ordered (xyz, bits of the cloud index) */
    std::vector<uint32_t> h_idx_of_pos; /* cloud index of the tree's p-th point */
    bool deep_tree = false;         /* built on the device (linear BVH): the kNN kernel takes its deep-stack form */
    tmx::dbuf<float> w_d2;          /* per tree point: squared distance to the current query */
    tmx::dbuf<uint32_t> w_sel;      /* compacted positions + count */
    tmx::dbuf<float> w_q;           /* query batch */
    tmx::dbuf<int32_t> w_idx;       /* kNN lists / gathered member indices */
    tmx::dbuf<float> w_out_d2;      /* kNN lists / gathered member distances */
    tmx::dbuf<uint32_t> w_cnt;
    std::mutex mutex; /* the reference guards its tree with a mutex (include/pointcloud:90): so does the handle */
};

namespace tmx {
inline ctx_guard guard_of(const tm_cloud* c) { return guard_of(c ? c->ctx : nullptr); }
}  // namespace tmx

namespace tmx {

constexpr int kCloudStack = 24;
constexpr int kCloudK = 32;       /* capacity of the in-LDS neighbour list */
constexpr int kCloudThreads = 128;

/* host: median-split kd-tree with a bounding box per node (pre-order numbering). lo.w / hi.w carry the links:
 * internal node: left / right child; leaf: begin | 0x80000000 / end into the leaf-ordered point array. Same
 * structure as the tree of the voxel / curvature builds (tm_model.hip), built here for an arbitrary subset. */
struct kd_node_builder {
    const std::vector<float4>* pos;
    std::vector<float4> lo, hi;
    std::vector<uint32_t> order;
    static float u2f(uint32_t u) {
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
    int build(int b, int e, int depth) {
        const int id = (int)lo.size();
        float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        for (int i = b; i < e; ++i) {
            const float4& p = (*pos)[order[i]];
            const float c[3] = {p.x, p.y, p.z};
            for (int d = 0; d < 3; ++d) {
                mn[d] = std::min(mn[d], c[d]);
                mx[d] = std::max(mx[d], c[d]);
            }
        }
        lo.push_back(make_float4(mn[0], mn[1], mn[2], u2f((uint32_t)b | 0x80000000u)));
        hi.push_back(make_float4(mx[0], mx[1], mx[2], u2f((uint32_t)e)));
        if (e - b <= 8 || depth >= kCloudStack - 2) return id;
        int dim = 0;
        if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
        if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        if (!(mx[dim] - mn[dim] > 0.f)) return id;
        const int mid = (b + e) / 2;
        auto coord = [&](uint32_t i) {
            const float4& p = (*pos)[i];
            return dim == 0 ? p.x : (dim == 1 ? p.y : p.z);
        };
        std::nth_element(order.begin() + b, order.begin() + mid, order.begin() + e,
                         [&](uint32_t a, uint32_t c) { return coord(a) < coord(c); });
        const int l = build(b, mid, depth + 1);
        const int r = build(mid, e, depth + 1);
        lo[id].w = u2f((uint32_t)l);
        hi[id].w = u2f((uint32_t)r);
        return id;
    }
};

__device__ __forceinline__ float cloud_box_d2(const float q[3], float4 lo, float4 hi) {
    float dx = fmaxf(fmaxf(lo.x - q[0], q[0] - hi.x), 0.f);
    float dy = fmaxf(fmaxf(lo.y - q[1], q[1] - hi.y), 0.f);
    float dz = fmaxf(fmaxf(lo.z - q[2], q[2] - hi.z), 0.f);
    return (dx * dx + dy * dy) + dz * dz;
}

/* k nearest tree points of every query, list sorted by (d2, cloud index). A subtree is skipped only when its
 * box distance EXCEEDS the current k-th distance (rounding is monotone, so the box bound never exceeds a
 * contained point's computed d2): equal-distance candidates with a lower index are always seen. */
template <int STACK>
__global__ __launch_bounds__(kCloudThreads) void k_cloud_knn(const float4* __restrict__ kd_lo,
                                                             const float4* __restrict__ kd_hi,
                                                             const float4* __restrict__ pts,
                                                             const float* __restrict__ queries, uint32_t nq, uint32_t k,
                                                             int32_t* __restrict__ out_idx, float* __restrict__ out_d2,
                                                             uint32_t* __restrict__ out_count) {
    /* STACK > 32: the device-built tree (deeper than a median kd-tree): node ids only, the box distance is taken
     * again at the pop (with the bounds the stack would not fit next to the neighbour lists) */
    constexpr bool kBounds = STACK <= 32;
    __shared__ uint32_t st_node[STACK][kCloudThreads];
    __shared__ float st_bound[kBounds ? STACK : 1][kCloudThreads];
    __shared__ float nb_d2[kCloudK][kCloudThreads];
    __shared__ uint32_t nb_idx[kCloudK][kCloudThreads];
    const uint32_t t = blockIdx.x * kCloudThreads + threadIdx.x;
    if (t >= nq) return;
    const uint32_t tx = threadIdx.x;
    const float q[3] = {queries[3 * (size_t)t], queries[3 * (size_t)t + 1], queries[3 * (size_t)t + 2]};
    uint32_t have = 0;
    const float inf = __builtin_inff();
    if (fabsf(q[0]) < inf && fabsf(q[1]) < inf && fabsf(q[2]) < inf) { /* a non-finite query has no neighbours */
        int sp = 1;
        st_node[0][tx] = 0u;
        if (kBounds) st_bound[0][tx] = 0.f;
        while (sp > 0) {
            --sp;
            uint32_t node = st_node[sp][tx];
            if (have == k && (kBounds ? st_bound[sp][tx] : cloud_box_d2(q, kd_lo[node], kd_hi[node])) > nb_d2[k - 1][tx]) continue;
            for (;;) {
                const uint32_t a = __float_as_uint(kd_lo[node].w), b = __float_as_uint(kd_hi[node].w);
                if (a & 0x80000000u) {
                    for (uint32_t p = a & 0x7FFFFFFFu; p < b; ++p) {
                        const float4 P = pts[p];
                        const float dx = P.x - q[0], dy = P.y - q[1], dz = P.z - q[2];
                        const float d2 = (dx * dx + dy * dy) + dz * dz;
                        const uint32_t idx = __float_as_uint(P.w);
                        if (have == k) {
                            const float wd = nb_d2[k - 1][tx];
                            const uint32_t wi = nb_idx[k - 1][tx];
                            if (!(d2 < wd || (d2 == wd && idx < wi))) continue;
                        }
                        int pos = (int)(have < k ? have : k - 1);
                        while (pos > 0) {
                            const float pd = nb_d2[pos - 1][tx];
                            const uint32_t pi = nb_idx[pos - 1][tx];
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
                const float dl = cloud_box_d2(q, kd_lo[a], kd_hi[a]);
                const float dr = cloud_box_d2(q, kd_lo[b], kd_hi[b]);
                const uint32_t nearc = dl <= dr ? a : b, farc = dl <= dr ? b : a;
                const float dn = fminf(dl, dr), df = fmaxf(dl, dr);
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
    }
    for (uint32_t j = 0; j < k; ++j) {
        out_idx[(size_t)t * k + j] = j < have ? (int32_t)nb_idx[j][tx] : -1;
        out_d2[(size_t)t * k + j] = j < have ? nb_d2[j][tx] : __builtin_inff();
    }
    out_count[t] = have;
}

/* the same lists with eight lanes per query (k_knn_groups, tm_knn.h; default): queries as packed xyz — a non-finite one
 * is not walked and keeps the empty list —, rows of k (index or -1, d2 or inf) and the number of neighbours found */
struct cloud_queries_xyz {
    const float* q;
    __device__ float4 operator()(uint32_t qi) const {
        const float x = q[3 * (size_t)qi], y = q[3 * (size_t)qi + 1], z = q[3 * (size_t)qi + 2];
        const float inf = __builtin_inff();
        const bool fin = fabsf(x) < inf && fabsf(y) < inf && fabsf(z) < inf;
        return make_float4(fin ? x : __builtin_nanf(""), y, z, 0.f);
    }
};
struct cloud_sink_rows {
    int32_t* idx;
    float* d2;
    uint32_t* count;
    uint32_t k, n_points;
    __device__ void operator()(uint32_t qi, float4 Q, uint32_t l, const float d[4], const uint32_t ix[4], float) const {
        for (uint32_t r = 0; r < 4u; ++r) {
            const uint32_t j = 4u * l + r;
            if (j < k) {
                idx[(size_t)qi * k + j] = (int32_t)ix[r]; /* an unused entry is 0xFFFFFFFF = -1 */
                d2[(size_t)qi * k + j] = d[r];
            }
        }
        if (l == 0u) count[qi] = Q.x == Q.x ? min(k, n_points) : 0u;
    }
};

/* a tree over a subset: the points' w = row of the subset list -> the cloud's own index */
__global__ __launch_bounds__(256) void k_cloud_remap_w(float4* __restrict__ pts, uint32_t n, const uint32_t* __restrict__ subset) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    float4 P = pts[t];
    if (P.x == P.x) {
        P.w = __uint_as_float(subset[__float_as_uint(P.w)]);
        pts[t] = P;
    }
}

/* squared distance of every tree point to one query (coalesced 16-B reads, 4-B writes) */
__global__ __launch_bounds__(256) void k_cloud_all_d2(const float4* __restrict__ pts, uint32_t n, float qx, float qy,
                                                      float qz, float* __restrict__ d2) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float4 P = pts[t];
    const float dx = P.x - qx, dy = P.y - qy, dz = P.z - qz;
    d2[t] = (dx * dx + dy * dy) + dz * dz;
}

/* (d2, cloud index) of the selected positions */
__global__ __launch_bounds__(256) void k_cloud_gather(const float4* __restrict__ pts, const float* __restrict__ d2,
                                                      const uint32_t* __restrict__ sel, uint32_t n_sel,
                                                      int32_t* __restrict__ out_idx, float* __restrict__ out_d2) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_sel) return;
    const uint32_t p = sel[t];
    out_idx[t] = (int32_t)__float_as_uint(pts[p].w);
    out_d2[t] = d2[p];
}

struct within_r2 {
    float r2;
    __host__ __device__ bool operator()(float d2) const { return d2 <= r2; }
};

}  // namespace tmx

using namespace tmx;

/* every tree point's squared distance to q on the device; members with d2 <= r2 compacted and brought back as
 * (d2, cloud index) pairs sorted ascending — the few survivors are ordered on the host */
static int cloud_members(tm_cloud* c, const float q[3], float r2, std::vector<std::pair<float, uint32_t>>& out) {
    out.clear();
    if (c->n_kd == 0) return TM_OK;
    tm_ctx* ctx = c->ctx;
    TM_FORWARD(c->w_d2.ensure(c->n_kd));
    TM_FORWARD(c->w_sel.ensure((size_t)c->n_kd + 1));
    hipLaunchKernelGGL(k_cloud_all_d2, dim3((c->n_kd + 255) / 256), dim3(256), 0, ctx->stream, c->kd_pts.p, c->n_kd,
                       q[0], q[1], q[2], c->w_d2.p);
    TM_HIP_CHECK(hipGetLastError());
    size_t tmp = 0;
    rocprim::counting_iterator<uint32_t> slots(0u);
    auto flags = rocprim::make_transform_iterator(c->w_d2.p, within_r2{r2});
    TM_HIP_CHECK(rocprim::select(nullptr, tmp, slots, flags, c->w_sel.p, c->w_sel.p + c->n_kd, (size_t)c->n_kd,
                                 ctx->stream));
    TM_FORWARD(scratch_ensure(ctx, tmp));
    TM_HIP_CHECK(rocprim::select(ctx->scratch.p, tmp, slots, flags, c->w_sel.p, c->w_sel.p + c->n_kd, (size_t)c->n_kd,
                                 ctx->stream));
    uint32_t cnt = 0;
    TM_HIP_CHECK(hipMemcpyAsync(&cnt, c->w_sel.p + c->n_kd, 4, hipMemcpyDeviceToHost, ctx->stream));
    TM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (cnt == 0) return TM_OK;
    TM_FORWARD(c->w_idx.ensure(cnt));
    TM_FORWARD(c->w_out_d2.ensure(cnt));
    hipLaunchKernelGGL(k_cloud_gather, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, c->kd_pts.p, c->w_d2.p,
                       c->w_sel.p, cnt, c->w_idx.p, c->w_out_d2.p);
    TM_HIP_CHECK(hipGetLastError());
    std::vector<int32_t> idx(cnt);
    std::vector<float> d2(cnt);
    TM_HIP_CHECK(hipMemcpyAsync(idx.data(), c->w_idx.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    TM_HIP_CHECK(hipMemcpyAsync(d2.data(), c->w_out_d2.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    TM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    out.resize(cnt);
    for (uint32_t k = 0; k < cnt; ++k) out[k] = {d2[k], (uint32_t)idx[k]};
    std::sort(out.begin(), out.end());
    return TM_OK;
}

extern "C" {

int tm_cloud_create(tm_ctx* ctx, const float* surfels, uint32_t n, const uint32_t* subset, uint32_t n_subset,
                    tm_cloud** out) {
    TM_GUARD(ctx);
    TM_REQUIRE(ctx && out && (surfels || n == 0), TM_ERR_INVALID, "tm_cloud_create: null argument");
    TM_REQUIRE(subset || n_subset == 0, TM_ERR_INVALID, "tm_cloud_create: null subset");
    *out = nullptr;
    for (uint32_t k = 0; k < n_subset; ++k)
        TM_REQUIRE(subset[k] < n, TM_ERR_INVALID, "tm_cloud_create: subset index out of range");
    TM_HIP_CHECK(hipSetDevice(ctx->device));
    tm_cloud* c = new tm_cloud();
    c->ctx = ctx;
    c->n = n;
    if (n) c->h_cloud.assign(surfels, surfels + (size_t)n * 12);
    if (!ctx->opt.model_host_build && n) {
        /* round 4: the tree is built on the device like every other tree of the library — the (subset of the) cloud goes
         * up raw, tmx::curve_order sorts it along the Hilbert curve, tmx::lbvh_build links the boxes; the host build
         * below (a recursive median split on one thread, ~1 s per million points) stays as the cross-check
         * (model_host_build = 1). The neighbour order of the queries does not depend on the tree: (d2, cloud index). */
        const uint32_t n_in = n_subset ? n_subset : n;
        std::vector<float> rows;
        if (n_subset) { /* set_indices(subset): the tree holds these points only */
            rows.resize((size_t)n_subset * 12);
            for (uint32_t k = 0; k < n_subset; ++k) memcpy(&rows[(size_t)k * 12], surfels + (size_t)subset[k] * 12, 48);
        }
        dbuf<float> d_cloud;
        dbuf<unsigned long long> d_keys;
        dbuf<uint32_t> d_order, d_cnt, d_sub;
        dbuf<unsigned char> d_bounds;
        dbuf<uint4> d_q;
        auto cleanup = [&]() {
            d_cloud.release();
            d_keys.release();
            d_order.release();
            d_cnt.release();
            d_sub.release();
            d_bounds.release();
            d_q.release();
        };
        int rc;
        if ((rc = d_cloud.alloc((size_t)n_in * 12)) || (rc = d_bounds.alloc(64)) || (rc = d_cnt.alloc(3 + kLbvhMaxDepth + 3 + 1)) ||
            (rc = d_q.alloc(2 * (size_t)n_in)) || (rc = c->kd_lo.alloc(2 * (size_t)n_in)) || (rc = c->kd_hi.alloc(2 * (size_t)n_in)) ||
            (rc = c->kd_pts.alloc(n_in)) || (n_subset && (rc = d_sub.alloc(n_subset)))) {
            cleanup();
            tm_cloud_destroy(c);
            return rc;
        }
        uint32_t* d_nfin = d_cnt.p + 3 + kLbvhMaxDepth + 3;
        uint32_t n_fin = 0;
        hipError_t e = hipMemcpyAsync(d_cloud.p, n_subset ? rows.data() : surfels, (size_t)n_in * 48, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && n_subset)
            e = hipMemcpyAsync(d_sub.p, subset, (size_t)n_subset * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_nfin, 0, 4, ctx->stream);
        if (e == hipSuccess) {
            rc = curve_order(ctx, d_cloud.p, n_in, true, d_bounds.p, d_keys, d_order);
            if (rc) {
                cleanup();
                tm_cloud_destroy(c);
                return rc;
            }
            curve_points(ctx, d_cloud.p, d_order.p, n_in, c->kd_pts.p, d_nfin);
            if (n_subset) /* w = row of the subset list -> the cloud's own index */
                hipLaunchKernelGGL(k_cloud_remap_w, dim3((n_in + 255) / 256), dim3(256), 0, ctx->stream, c->kd_pts.p, n_in, d_sub.p);
            e = hipMemcpyAsync(&n_fin, d_nfin, 4, hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && n_fin > 0) {
            lbvh_build(ctx, d_keys.p, c->kd_pts.p, n_fin, c->kd_lo.p, c->kd_hi.p, d_q.p, d_q.p + n_in, d_cnt.p);
            std::vector<float4> h_pts(n_fin);
            e = hipMemcpyAsync(h_pts.data(), c->kd_pts.p, (size_t)n_fin * 16, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            c->h_idx_of_pos.resize(n_fin);
            for (uint32_t k = 0; k < n_fin; ++k) memcpy(&c->h_idx_of_pos[k], &h_pts[k].w, 4);
        }
        cleanup();
        if (e != hipSuccess) {
            set_error("tm_cloud_create: %s", hipGetErrorString(e));
            tm_cloud_destroy(c);
            return TM_ERR_HIP;
        }
        c->n_kd = n_fin;
        c->deep_tree = true;
        *out = c;
        return TM_OK;
    }
    std::vector<float4> pos(n);
    for (uint32_t i = 0; i < n; ++i) {
        const float* s = surfels + (size_t)i * 12;
        pos[i] = make_float4(s[0], s[1], s[2], 0.f);
    }
    kd_node_builder kb;
    kb.pos = &pos;
    auto finite = [&](uint32_t i) {
        return std::isfinite(pos[i].x) && std::isfinite(pos[i].y) && std::isfinite(pos[i].z);
    };
    if (n_subset) { /* set_indices(subset): the tree holds these points only (include/impl/pointcloud.hpp:231-234) */
        for (uint32_t k = 0; k < n_subset; ++k)
            if (finite(subset[k])) kb.order.push_back(subset[k]);
    } else {
        for (uint32_t i = 0; i < n; ++i)
            if (finite(i)) kb.order.push_back(i);
    }
    c->n_kd = (uint32_t)kb.order.size();
    if (c->n_kd) {
        kb.build(0, (int)c->n_kd, 0);
        std::vector<float4> kd_pts(c->n_kd);
        for (uint32_t k = 0; k < c->n_kd; ++k) {
            const float4& p = pos[kb.order[k]];
            kd_pts[k] = make_float4(p.x, p.y, p.z, kd_node_builder::u2f(kb.order[k]));
        }
        c->h_idx_of_pos = kb.order;
        int rc;
        if ((rc = c->kd_lo.alloc(kb.lo.size())) || (rc = c->kd_hi.alloc(kb.hi.size())) ||
            (rc = c->kd_pts.alloc(kd_pts.size()))) {
            tm_cloud_destroy(c);
            return rc;
        }
        hipError_t e = hipMemcpyAsync(c->kd_lo.p, kb.lo.data(), kb.lo.size() * 16, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(c->kd_hi.p, kb.hi.data(), kb.hi.size() * 16, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(c->kd_pts.p, kd_pts.data(), kd_pts.size() * 16, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            set_error("tm_cloud_create: %s", hipGetErrorString(e));
            tm_cloud_destroy(c);
            return TM_ERR_HIP;
        }
    }
    *out = c;
    return TM_OK;
}

int tm_cloud_destroy(tm_cloud* c) {
    if (!c) return TM_OK;
    tmx::ctx_guard tm_guard_;
    if (ctx_alive(c->ctx)) {
        tm_guard_ = tmx::guard_of(c->ctx);
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        c->kd_lo.release();
        c->kd_hi.release();
        c->kd_pts.release();
        c->w_d2.release();
        c->w_sel.release();
        c->w_q.release();
        c->w_idx.release();
        c->w_out_d2.release();
        c->w_cnt.release();
    }
    delete c;
    return TM_OK;
}

uint32_t tm_cloud_tree_size(const tm_cloud* c) {
    TM_GUARD(c); return c ? c->n_kd : 0u; }

int tm_cloud_knn(tm_cloud* c, const float* queries_xyz, uint32_t n_queries, uint32_t k, int32_t* indices,
                 float* sq_dists, uint32_t* counts) {
    TM_GUARD(c);
    TM_REQUIRE(c && (queries_xyz || n_queries == 0) && indices && sq_dists, TM_ERR_INVALID,
               "tm_cloud_knn: null argument");
    TM_REQUIRE(k >= 1, TM_ERR_INVALID, "tm_cloud_knn: k must be at least 1");
    if (n_queries == 0) return TM_OK;
    std::lock_guard<std::mutex> lk(c->mutex);
    tm_ctx* ctx = c->ctx;
    TM_HIP_CHECK(hipSetDevice(ctx->device));
    if (c->n_kd == 0) {
        for (size_t t = 0; t < (size_t)n_queries * k; ++t) {
            indices[t] = -1;
            sq_dists[t] = std::numeric_limits<float>::infinity();
        }
        if (counts) std::fill(counts, counts + n_queries, 0u);
        return TM_OK;
    }
    if (k <= (uint32_t)kCloudK) { /* the batch in one launch */
        TM_FORWARD(c->w_q.ensure((size_t)n_queries * 3));
        TM_FORWARD(c->w_idx.ensure((size_t)n_queries * k));
        TM_FORWARD(c->w_out_d2.ensure((size_t)n_queries * k));
        TM_FORWARD(c->w_cnt.ensure(n_queries));
        TM_HIP_CHECK(hipMemcpyAsync(c->w_q.p, queries_xyz, (size_t)n_queries * 12, hipMemcpyHostToDevice, ctx->stream));
        if (ctx->opt.curv_groups) {
            const cloud_queries_xyz qs{c->w_q.p};
            const cloud_sink_rows sink{c->w_idx.p, c->w_out_d2.p, c->w_cnt.p, k, c->n_kd};
            const dim3 grid((n_queries + (uint32_t)kKnnGroupsPerBlock - 1u) / (uint32_t)kKnnGroupsPerBlock);
            if (c->deep_tree)
                hipLaunchKernelGGL((k_knn_groups<kLbvhStack, cloud_queries_xyz, cloud_sink_rows>), grid, dim3(256), 0, ctx->stream,
                                   c->kd_lo.p, c->kd_hi.p, c->kd_pts.p, k, qs, n_queries, sink);
            else
                hipLaunchKernelGGL((k_knn_groups<kCloudStack, cloud_queries_xyz, cloud_sink_rows>), grid, dim3(256), 0, ctx->stream,
                                   c->kd_lo.p, c->kd_hi.p, c->kd_pts.p, k, qs, n_queries, sink);
        } else { /* one thread per query, lists in LDS */
            hipLaunchKernelGGL(c->deep_tree ? k_cloud_knn<kLbvhStack> : k_cloud_knn<kCloudStack>,
                               dim3((n_queries + kCloudThreads - 1) / kCloudThreads), dim3(kCloudThreads), 0, ctx->stream,
                               c->kd_lo.p, c->kd_hi.p, c->kd_pts.p, c->w_q.p, n_queries, k, c->w_idx.p, c->w_out_d2.p,
                               c->w_cnt.p);
        }
        TM_HIP_CHECK(hipGetLastError());
        TM_HIP_CHECK(hipMemcpyAsync(indices, c->w_idx.p, (size_t)n_queries * k * 4, hipMemcpyDeviceToHost, ctx->stream));
        TM_HIP_CHECK(hipMemcpyAsync(sq_dists, c->w_out_d2.p, (size_t)n_queries * k * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (counts)
            TM_HIP_CHECK(hipMemcpyAsync(counts, c->w_cnt.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost, ctx->stream));
        TM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return TM_OK;
    }
    /* long lists: all squared distances from one pass over the points, the k smallest picked on the host */
    TM_FORWARD(c->w_d2.ensure(c->n_kd));
    std::vector<float> d2(c->n_kd);
    std::vector<std::pair<float, uint32_t>> nb;
    for (uint32_t t = 0; t < n_queries; ++t) {
        const float* q = queries_xyz + 3 * (size_t)t;
        uint32_t have = 0;
        if (std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2])) {
            hipLaunchKernelGGL(k_cloud_all_d2, dim3((c->n_kd + 255) / 256), dim3(256), 0, ctx->stream, c->kd_pts.p, c->n_kd,
                               q[0], q[1], q[2], c->w_d2.p);
            TM_HIP_CHECK(hipGetLastError());
            TM_HIP_CHECK(hipMemcpyAsync(d2.data(), c->w_d2.p, (size_t)c->n_kd * 4, hipMemcpyDeviceToHost, ctx->stream));
            TM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            nb.resize(c->n_kd);
            for (uint32_t p = 0; p < c->n_kd; ++p) nb[p] = {d2[p], c->h_idx_of_pos[p]};
            have = std::min<uint32_t>(k, c->n_kd);
            std::partial_sort(nb.begin(), nb.begin() + have, nb.end());
        }
        for (uint32_t j = 0; j < k; ++j) {
            indices[(size_t)t * k + j] = j < have ? (int32_t)nb[j].second : -1;
            sq_dists[(size_t)t * k + j] = j < have ? nb[j].first : std::numeric_limits<float>::infinity();
        }
        if (counts) counts[t] = have;
    }
    return TM_OK;
}

int tm_cloud_radius_search(tm_cloud* c, const float query_xyz[3], float radius, int32_t* indices, float* sq_dists,
                           uint32_t cap, uint32_t* count) {
    TM_GUARD(c);
    TM_REQUIRE(c && query_xyz && count, TM_ERR_INVALID, "tm_cloud_radius_search: null argument");
    *count = 0;
    if (!(radius >= 0.f) || c->n_kd == 0) return TM_OK;
    if (!(std::isfinite(query_xyz[0]) && std::isfinite(query_xyz[1]) && std::isfinite(query_xyz[2]))) return TM_OK;
    std::lock_guard<std::mutex> lk(c->mutex);
    TM_HIP_CHECK(hipSetDevice(c->ctx->device));
    std::vector<std::pair<float, uint32_t>> nb;
    TM_FORWARD(cloud_members(c, query_xyz, radius * radius, nb));
    *count = (uint32_t)nb.size();
    const uint32_t m = std::min<uint32_t>(*count, cap);
    for (uint32_t k = 0; k < m; ++k) {
        if (indices) indices[k] = (int32_t)nb[k].second;
        if (sq_dists) sq_dists[k] = nb[k].first;
    }
    if ((indices || sq_dists) && *count > cap) {
        set_error("tm_cloud_radius_search: %u neighbours, room for %u", *count, cap);
        return TM_ERR_CAPACITY;
    }
    return TM_OK;
}

/* principal_curvatures(cloud, idx, neighbours) of the reference (include/impl/pointcloud.hpp:3-44) over an
 * explicit neighbour list, as curvature(float r, idx) feeds it (:206-209). Same arithmetic as k_curvature. */
int tm_cloud_curvature_of(const tm_cloud* c, uint32_t idx, const int32_t* neighbours, uint32_t n_neighbours,
                          float curv5[5]) {
    TM_GUARD(c);
    TM_REQUIRE(c && curv5 && (neighbours || n_neighbours == 0), TM_ERR_INVALID, "tm_cloud_curvature_of: null argument");
    TM_REQUIRE(idx < c->n, TM_ERR_INVALID, "tm_cloud_curvature_of: index out of range");
    for (int k = 0; k < 5; ++k) curv5[k] = 0.f;
    if (n_neighbours == 0) return TM_OK;
    for (uint32_t j = 0; j < n_neighbours; ++j)
        TM_REQUIRE(neighbours[j] >= 0 && (uint32_t)neighbours[j] < c->n, TM_ERR_INVALID,
                   "tm_cloud_curvature_of: neighbour out of range");
    const float* s0 = c->h_cloud.data() + (size_t)idx * 12;
    const float n0[3] = {s0[4], s0[5], s0[6]};
    float M[3][3];
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) M[r][cc] = (r == cc ? 1.f : 0.f) - n0[r] * n0[cc];
    float cen[3] = {0.f, 0.f, 0.f};
    for (uint32_t j = 0; j < n_neighbours; ++j) {
        const float* sj = c->h_cloud.data() + (size_t)neighbours[j] * 12;
        for (int r = 0; r < 3; ++r) {
            const float pr = (M[r][0] * sj[4] + M[r][1] * sj[5]) + M[r][2] * sj[6];
            cen[r] = cen[r] + (pr - cen[r]) / (float)(j + 1);
        }
    }
    float cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (uint32_t j = 0; j < n_neighbours; ++j) {
        const float* sj = c->h_cloud.data() + (size_t)neighbours[j] * 12;
        float d[3];
        for (int r = 0; r < 3; ++r) d[r] = ((M[r][0] * sj[4] + M[r][1] * sj[5]) + M[r][2] * sj[6]) - cen[r];
        cov[0] += d[0] * d[0];
        cov[1] += d[0] * d[1];
        cov[2] += d[0] * d[2];
        cov[3] += d[1] * d[1];
        cov[4] += d[1] * d[2];
        cov[5] += d[2] * d[2];
    }
    float ev[3], vmax[3];
    jacobi_eigen3(cov, ev, vmax);
    const float area_inv = 1.0f / (float)n_neighbours;
    curv5[0] = vmax[0];
    curv5[1] = vmax[1];
    curv5[2] = vmax[2];
    curv5[3] = ev[1] * area_inv;
    curv5[4] = ev[2] * area_inv;
    return TM_OK;
}

}  // extern "C"
