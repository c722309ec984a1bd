/*
 * tm_scene_build.hip — device-side preparation of a scene for the scoring kernels.
 *
 * The reference keeps the scene as a pcl::PointCloud plus two host bit-vectors (tangent_mask_, mask_,
 * include/impl/scene.hpp:41-58,87-90) and walks it point by point. Here the cloud is uploaded once
 * (48 B per surfel) and everything the kernels read is derived on the GPU:
 *
 *   k_scene_original   opos / otan / oref in original order (hypothesis generation addresses samples
 *                      by their original index)
 *   k_scene_bounds     bounding box of the finite points (ordered-integer atomics)
 *   k_scene_morton     63-bit Hilbert-curve index per point (TM_SCENE_MORTON=1: plain Morton code);
 *                      non-finite points sort last. Hilbert order has no long jumps, so the 64-point
 *                      tiles are tighter: 12 % fewer tiles survive culling than with Morton order
 *   rocprim radix sort (code, index) pairs -> `order` (sorted slot -> original index); stable, so
 *                      equal codes keep ascending index
 *   k_scene_gather     spos / spos3 / sref in sorted order; masked, non-finite and padding slots carry
 *                      NaN in x so that they fail the in-grid test; counts the live points
 *   k_scene_tiles      bounding sphere of every 64 consecutive sorted points (one wave per tile)
 *   k_scene_subs       bounding spheres of the four 16-point quarters of every tile
 *   k_scene_supers     bounding sphere of every 4096 consecutive sorted points (one block each), and of
 *                      every 512 (the chunks of the grouped scoring kernel)
 *
 * The order does not depend on mask_, so tm_scene_set_mask re-runs only the last four steps. Inlier
 * counts and fixed-point scores are independent of the order of the points, and the spheres are
 * conservative (radius rounded up), so none of this changes a result; it only decides what the culled
 * kernel may skip. All of it is HBM-bound streaming: ~0.2 KB per point in total.
 */
#include <cstring>

#include "tm_internal.h"
#include "tm_score.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>

namespace tmx {

struct scene_bounds {
    unsigned int lo[3]; /* ordered-integer encodings of the minimum / maximum finite coordinate */
    unsigned int hi[3];
    unsigned int cmax;  /* bits of max |coordinate| (non-negative floats order like their bits) */
    unsigned int live;  /* points that are finite and not masked */
};

__device__ __forceinline__ unsigned int f2ord(float f) {
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(unsigned int u) {
    unsigned int b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
__device__ __forceinline__ bool finite3(float x, float y, float z) {
    return isfinite(x) && isfinite(y) && isfinite(z);
}

__global__ __launch_bounds__(256) void k_scene_original(const float* __restrict__ cloud,
                                                        const int32_t* __restrict__ tmask,
                                                        const int32_t* __restrict__ mask, uint32_t n,
                                                        float4* __restrict__ opos, float4* __restrict__ otan,
                                                        float4* __restrict__ oref) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4* s = reinterpret_cast<const float4*>(cloud) + (size_t)i * 3;
    float4 a = s[0], b = s[1], c = s[2];
    bool ut = tmask[i] != 0;
    opos[i] = make_float4(a.x, a.y, a.z, ut ? 1.f : 0.f);
    otan[i] = make_float4(c.y, c.z, c.w, mask[i] != 0 ? 1.f : 0.f);
    oref[i] = ut ? make_float4(c.y, c.z, c.w, 1.f) : make_float4(b.x, b.y, b.z, 0.f);
}

__global__ __launch_bounds__(256) void k_scene_bounds(const float* __restrict__ cloud, uint32_t n,
                                                      scene_bounds* __restrict__ out) {
    __shared__ unsigned int s_lo[3][4], s_hi[3][4], s_c[4];
    unsigned int lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0u, 0u, 0u}, cm = 0u;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float4 a = reinterpret_cast<const float4*>(cloud)[(size_t)i * 3];
        if (!finite3(a.x, a.y, a.z)) continue;
        const float q[3] = {a.x, a.y, a.z};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            unsigned int o = f2ord(q[d]);
            lo[d] = min(lo[d], o);
            hi[d] = max(hi[d], o);
            cm = max(cm, __float_as_uint(fabsf(q[d])));
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = min(lo[d], (unsigned int)__shfl_xor((int)lo[d], off));
            hi[d] = max(hi[d], (unsigned int)__shfl_xor((int)hi[d], off));
        }
        cm = max(cm, (unsigned int)__shfl_xor((int)cm, off));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int d = 0; d < 3; ++d) {
            s_lo[d][w] = lo[d];
            s_hi[d][w] = hi[d];
        }
        s_c[w] = cm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int d = 0; d < 3; ++d) {
            atomicMin(&out->lo[d], min(min(s_lo[d][0], s_lo[d][1]), min(s_lo[d][2], s_lo[d][3])));
            atomicMax(&out->hi[d], max(max(s_hi[d][0], s_hi[d][1]), max(s_hi[d][2], s_hi[d][3])));
        }
        atomicMax(&out->cmax, max(max(s_c[0], s_c[1]), max(s_c[2], s_c[3])));
    }
}

__device__ __forceinline__ unsigned long long spread21_dev(unsigned long long v) {
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ __launch_bounds__(256) void k_scene_morton(const float* __restrict__ cloud, uint32_t n,
                                                      const scene_bounds* __restrict__ bnd, int hilbert,
                                                      unsigned long long* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 a = reinterpret_cast<const float4*>(cloud)[(size_t)i * 3];
    unsigned long long code = ~0ull;
    if (finite3(a.x, a.y, a.z)) {
        const double mn0 = ord2f(bnd->lo[0]), mn1 = ord2f(bnd->lo[1]), mn2 = ord2f(bnd->lo[2]);
        const double e0 = (double)ord2f(bnd->hi[0]) - mn0, e1 = (double)ord2f(bnd->hi[1]) - mn1,
                     e2 = (double)ord2f(bnd->hi[2]) - mn2;
        const double ext = fmax(fmax(e0, e1), e2);
        const double inv = ext > 0.0 ? 2097151.0 / ext : 0.0;
        unsigned long long ix = (unsigned long long)(((double)a.x - mn0) * inv);
        unsigned long long iy = (unsigned long long)(((double)a.y - mn1) * inv);
        unsigned long long iz = (unsigned long long)(((double)a.z - mn2) * inv);
        if (hilbert) {
            unsigned int X[3] = {(unsigned int)ix, (unsigned int)iy, (unsigned int)iz};
            hilbert_transpose21(X);
            code = (spread21_dev(X[0]) << 2) | (spread21_dev(X[1]) << 1) | spread21_dev(X[2]);
        } else {
            code = spread21_dev(ix) | (spread21_dev(iy) << 1) | (spread21_dev(iz) << 2);
        }
    }
    keys[i] = code;
    vals[i] = i;
}

/* live points of a 256-thread workgroup -> one atomic (one per wave on the same word serialises: 16k of them cost
 * more than the pass over the points) */
__device__ __forceinline__ void block_count_live(bool live, unsigned int* total) {
    __shared__ unsigned int s_n[4];
    const unsigned long long b = __ballot(live);
    if ((threadIdx.x & 63) == 0) s_n[threadIdx.x >> 6] = (unsigned int)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = (s_n[0] + s_n[1]) + (s_n[2] + s_n[3]);
        if (t) atomicAdd(total, t);
    }
}

__global__ __launch_bounds__(256) void k_scene_gather(const float* __restrict__ cloud,
                                                      const int32_t* __restrict__ tmask,
                                                      const int32_t* __restrict__ mask,
                                                      const uint32_t* __restrict__ order, uint32_t n,
                                                      uint32_t n_pad, float4* __restrict__ spos,
                                                      float* __restrict__ spos3, float4* __restrict__ sref,
                                                      scene_bounds* __restrict__ bnd) {
    uint32_t k = blockIdx.x * 256 + threadIdx.x;
    bool live = false;
    if (k < n_pad) {
        const float nanv = __builtin_nanf("");
        float4 p = make_float4(nanv, 0.f, 0.f, 0.f), r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < n) {
            uint32_t i = order[k];
            const float4* s = reinterpret_cast<const float4*>(cloud) + (size_t)i * 3;
            float4 a = s[0], b = s[1], c = s[2];
            live = finite3(a.x, a.y, a.z) && mask[i] == 0;
            p = make_float4(live ? a.x : nanv, a.y, a.z, __uint_as_float(i));
            r = tmask[i] != 0 ? make_float4(c.y, c.z, c.w, 1.f) : make_float4(b.x, b.y, b.z, 0.f);
        }
        spos[k] = p;
        sref[k] = r;
        spos3[3 * (size_t)k] = p.x;
        spos3[3 * (size_t)k + 1] = p.y;
        spos3[3 * (size_t)k + 2] = p.z;
    }
    block_count_live(live, &bnd->live);
}

/* tm_scene_set_mask on a scene that is already sorted: only x of a point (NaN = not live) depends on mask_, so the
 * sorted arrays are patched in place — one coalesced pass over spos and a 4-byte look-up of the mask per point
 * instead of re-gathering 48-byte surfels through the order (0.17 ms per million points, twice per search). Same
 * result as k_scene_gather: live = finite and unmasked, y / z / index / sref do not change. */
__global__ __launch_bounds__(256) void k_scene_remask(const float* __restrict__ cloud,
                                                      const int32_t* __restrict__ mask, uint32_t n,
                                                      float4* __restrict__ spos, float* __restrict__ spos3,
                                                      scene_bounds* __restrict__ bnd) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    bool live = false;
    if (k < n) {
        float4 p = spos[k];
        const uint32_t i = __float_as_uint(p.w);
        const bool was_live = p.x == p.x;
        float x = p.x;
        if (mask[i] != 0)
            live = false;
        else if (was_live)
            live = true;
        else {
            x = cloud[12 * (size_t)i];
            live = finite3(x, p.y, p.z);
        }
        if (live != was_live) {
            p.x = live ? x : __builtin_nanf("");
            spos[k] = p;
            spos3[3 * (size_t)k] = p.x;
        }
    }
    block_count_live(live, &bnd->live);
}

__device__ __forceinline__ double wave_min(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

/* sphere = (centre of the bounding box of the live points, largest distance to it rounded up);
 * w < 0 marks a tile without live points */
__device__ __forceinline__ float4 sphere_of(double lo0, double lo1, double lo2, double hi0, double hi1,
                                            double hi2, float& cx, float& cy, float& cz) {
    cx = (float)(0.5 * (lo0 + hi0));
    cy = (float)(0.5 * (lo1 + hi1));
    cz = (float)(0.5 * (lo2 + hi2));
    return make_float4(cx, cy, cz, 0.f);
}
__device__ __forceinline__ float radius_up(double r2) {
    float r = (float)(sqrt(r2) * (1.0 + 1e-6));
    return nextafterf(r, __builtin_inff());
}

__global__ __launch_bounds__(256) void k_scene_tiles(const float4* __restrict__ spos, uint32_t n_tiles,
                                                     float4* __restrict__ tiles) {
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tiles) return; /* whole waves leave together */
    float4 p = spos[(size_t)t * 64 + (threadIdx.x & 63)];
    const bool live = p.x == p.x;
    if (__ballot(live) == 0ull) {
        if ((threadIdx.x & 63) == 0) tiles[t] = make_float4(0.f, 0.f, 0.f, -1.f);
        return;
    }
    const double big = 1e300;
    double lo0 = wave_min(live ? (double)p.x : big), hi0 = wave_max(live ? (double)p.x : -big);
    double lo1 = wave_min(live ? (double)p.y : big), hi1 = wave_max(live ? (double)p.y : -big);
    double lo2 = wave_min(live ? (double)p.z : big), hi2 = wave_max(live ? (double)p.z : -big);
    float cx, cy, cz;
    float4 s = sphere_of(lo0, lo1, lo2, hi0, hi1, hi2, cx, cy, cz);
    double dx = (double)p.x - cx, dy = (double)p.y - cy, dz = (double)p.z - cz;
    double r2 = wave_max(live ? dx * dx + dy * dy + dz * dz : 0.0);
    s.w = radius_up(r2);
    if ((threadIdx.x & 63) == 0) tiles[t] = s;
}

/* bounding spheres of the four 16-point quarters of every tile: 16-lane groups of one wave per tile */
__device__ __forceinline__ double group16_min(double v) {
    for (int off = 8; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double group16_max(double v) {
    for (int off = 8; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}
__global__ __launch_bounds__(256) void k_scene_subs(const float4* __restrict__ spos, uint32_t n_tiles,
                                                    float4* __restrict__ subs) {
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const uint32_t lane = threadIdx.x & 63;
    float4 p = spos[(size_t)t * 64 + lane];
    const bool live = p.x == p.x;
    const double big = 1e300;
    const double any = group16_max(live ? 1.0 : 0.0);
    double lo0 = group16_min(live ? (double)p.x : big), hi0 = group16_max(live ? (double)p.x : -big);
    double lo1 = group16_min(live ? (double)p.y : big), hi1 = group16_max(live ? (double)p.y : -big);
    double lo2 = group16_min(live ? (double)p.z : big), hi2 = group16_max(live ? (double)p.z : -big);
    float cx, cy, cz;
    float4 s = sphere_of(lo0, lo1, lo2, hi0, hi1, hi2, cx, cy, cz);
    double dx = (double)p.x - cx, dy = (double)p.y - cy, dz = (double)p.z - cz;
    double r2 = group16_max(live ? dx * dx + dy * dy + dz * dz : 0.0);
    s.w = radius_up(r2);
    if (any == 0.0) s = make_float4(0.f, 0.f, 0.f, -1.f);
    if ((lane & 15) == 0) subs[(size_t)t * 4 + (lane >> 4)] = s;
}

__global__ __launch_bounds__(256) void k_scene_supers(const float4* __restrict__ spos, uint32_t n_pad,
                                                      uint32_t span, float4* __restrict__ supers) {
    __shared__ double s_lo[3][4], s_hi[3][4], s_r[4];
    __shared__ int s_live[4];
    const uint32_t k0 = blockIdx.x * span, k1 = min(k0 + span, n_pad);
    const double big = 1e300;
    double lo[3] = {big, big, big}, hi[3] = {-big, -big, -big};
    int live = 0;
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += 256) {
        float4 p = spos[k];
        if (!(p.x == p.x)) continue;
        const double q[3] = {(double)p.x, (double)p.y, (double)p.z};
        for (int d = 0; d < 3; ++d) {
            lo[d] = fmin(lo[d], q[d]);
            hi[d] = fmax(hi[d], q[d]);
        }
        live = 1;
    }
    const int w = threadIdx.x >> 6;
    for (int d = 0; d < 3; ++d) {
        lo[d] = wave_min(lo[d]);
        hi[d] = wave_max(hi[d]);
    }
    unsigned long long any = __ballot(live != 0);
    if ((threadIdx.x & 63) == 0) {
        for (int d = 0; d < 3; ++d) {
            s_lo[d][w] = lo[d];
            s_hi[d][w] = hi[d];
        }
        s_live[w] = any != 0ull;
    }
    __syncthreads();
    if (!(s_live[0] | s_live[1] | s_live[2] | s_live[3])) {
        if (threadIdx.x == 0) supers[blockIdx.x] = make_float4(0.f, 0.f, 0.f, -1.f);
        return;
    }
    for (int d = 0; d < 3; ++d) {
        lo[d] = fmin(fmin(s_lo[d][0], s_lo[d][1]), fmin(s_lo[d][2], s_lo[d][3]));
        hi[d] = fmax(fmax(s_hi[d][0], s_hi[d][1]), fmax(s_hi[d][2], s_hi[d][3]));
    }
    float cx, cy, cz;
    float4 s = sphere_of(lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], cx, cy, cz);
    double r2 = 0.0;
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += 256) {
        float4 p = spos[k];
        if (!(p.x == p.x)) continue;
        double dx = (double)p.x - cx, dy = (double)p.y - cy, dz = (double)p.z - cz;
        r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
    }
    r2 = wave_max(r2);
    if ((threadIdx.x & 63) == 0) s_r[w] = r2;
    __syncthreads();
    if (threadIdx.x == 0) {
        s.w = radius_up(fmax(fmax(s_r[0], s_r[1]), fmax(s_r[2], s_r[3])));
        supers[blockIdx.x] = s;
    }
}

/* radius_search_inclusive membership: one bit per point (original order), one 64-bit word per wave */
__global__ __launch_bounds__(256) void k_radius_mask(const float4* __restrict__ opos, uint32_t n, uint32_t center,
                                                     float r2, unsigned long long* __restrict__ words) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const float4 c = opos[center];
    bool in = false;
    if (i < n) {
        float4 p = opos[i];
        in = len2(make3(p.x, p.y, p.z) - make3(c.x, c.y, c.z)) <= r2;
    }
    unsigned long long b = __ballot(in);
    if ((threadIdx.x & 63) == 0) words[i >> 6] = b;
}

/* uploads the cloud and the two masks (original order) */
int scene_upload(tm_scene* s, const float* surfels) {
    tm_ctx* c = s->ctx;
    const uint32_t n = s->n;
    TM_FORWARD(s->d_cloud.ensure((size_t)n * 12));
    TM_FORWARD(s->d_tmask.ensure(n));
    TM_FORWARD(s->d_mask.ensure(n));
    /* surfels = the caller's buffer (tm_scene_create); null = the cloud is on the device already */
    if (surfels)
        TM_HIP_CHECK(hipMemcpyAsync(s->d_cloud.p, surfels, (size_t)n * 48, hipMemcpyHostToDevice, c->stream));
    /* an all-zero mask (every new scene has two) is set on the device, not sent */
    const tmx::mask_mirror &ht = s->h_tmask, &hm = s->h_mask;
    if (ht.all_zero()) TM_HIP_CHECK(hipMemsetAsync(s->d_tmask.p, 0, (size_t)n * 4, c->stream));
    else TM_HIP_CHECK(hipMemcpyAsync(s->d_tmask.p, ht.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if (hm.all_zero()) TM_HIP_CHECK(hipMemsetAsync(s->d_mask.p, 0, (size_t)n * 4, c->stream));
    else TM_HIP_CHECK(hipMemcpyAsync(s->d_mask.p, hm.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    return TM_OK;
}

/* Curve keys of a raw cloud (48-byte surfels on the device) and the order they put the points in: keys_out[k] =
 * 63-bit Hilbert (or Morton) key of the k-th point along the curve, ascending; order_out[k] = its original index;
 * equal keys keep their index order (stable radix sort), non-finite points come last (key ~0). *bnd receives the
 * bounding box of the finite points, the largest |coordinate| and a zeroed live counter. Used for scenes and, since
 * round 4, for models (tm_model_create). Queued on the context's stream; synchronises once, after the sort. */
int curve_order(tm_ctx* c, const float* d_cloud, uint32_t n, bool hilbert, void* d_bounds,
                dbuf<unsigned long long>& keys_out, dbuf<uint32_t>& order_out) {
    scene_bounds* bnd = reinterpret_cast<scene_bounds*>(d_bounds);
    const dim3 blk(256), grid_n((n + 255) / 256);
    scene_bounds init;
    for (int d = 0; d < 3; ++d) {
        init.lo[d] = ~0u;
        init.hi[d] = 0u;
    }
    init.cmax = 0u;
    init.live = 0u;
    TM_HIP_CHECK(hipMemcpyAsync(bnd, &init, sizeof init, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_scene_bounds, dim3(std::min<uint32_t>(512u, (n + 255) / 256)), blk, 0, c->stream, d_cloud, n, bnd);
    dbuf<unsigned long long> k_in;
    dbuf<uint32_t> v_in;
    auto cleanup = [&]() {
        k_in.release();
        v_in.release();
    };
    int rc;
    if ((rc = k_in.alloc(n)) || (rc = keys_out.ensure(n)) || (rc = v_in.alloc(n)) || (rc = order_out.ensure(n))) {
        cleanup();
        return rc;
    }
    hipLaunchKernelGGL(k_scene_morton, grid_n, blk, 0, c->stream, d_cloud, n, bnd, hilbert ? 1 : 0, k_in.p, v_in.p);
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, k_in.p, keys_out.p, v_in.p, order_out.p, (size_t)n, 0, 64,
                                             c->stream);
    if (e == hipSuccess && (rc = scratch_ensure(c, tmp))) {
        cleanup();
        return rc;
    }
    if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(c->scratch.p, tmp, k_in.p, keys_out.p, v_in.p, order_out.p, (size_t)n, 0, 64,
                                      c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) {
        set_error("curve_order: sort failed: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

/* (re)derives every device array of the scene from d_cloud / d_tmask / d_mask. `resort` = the cloud
 * changed (new scene); false = only mask_ changed. */
int scene_build(tm_scene* s, bool resort) {
    tm_ctx* c = s->ctx;
    const uint32_t n = s->n;
    const uint32_t n_pad = ((n + kScoreTile - 1) / kScoreTile) * kScoreTile;
    const uint32_t n_tiles = n_pad / 64, n_super = (n_tiles + 63) / 64;
    const dim3 blk(256), grid_n((n + 255) / 256);
    TM_FORWARD(s->opos.ensure(n));
    TM_FORWARD(s->otan.ensure(n));
    TM_FORWARD(s->oref.ensure(n));
    TM_FORWARD(s->d_bounds.ensure(sizeof(scene_bounds)));
    TM_FORWARD(s->d_order.ensure(n));
    TM_FORWARD(s->spos.ensure(n_pad));
    TM_FORWARD(s->sref.ensure(n_pad));
    TM_FORWARD(s->spos3.ensure((size_t)n_pad * 3));
    TM_FORWARD(s->tiles.ensure(n_tiles));
    TM_FORWARD(s->supers.ensure(n_super));
    TM_FORWARD(s->subs.ensure((size_t)n_tiles * 4));
    TM_FORWARD(s->chunks.ensure(n_pad / 512));
    scene_bounds* bnd = reinterpret_cast<scene_bounds*>(s->d_bounds.p);
    hipLaunchKernelGGL(k_scene_original, grid_n, blk, 0, c->stream, s->d_cloud.p, s->d_tmask.p, s->d_mask.p, n,
                       s->opos.p, s->otan.p, s->oref.p);
    if (resort) {
        TM_FORWARD(curve_order(c, s->d_cloud.p, n, true, bnd, s->d_keys, s->d_order));
    } else {
        TM_HIP_CHECK(hipMemsetAsync(&bnd->live, 0, 4, c->stream));
    }
    if (!resort && s->n_sorted == n_pad)
        hipLaunchKernelGGL(k_scene_remask, dim3(n_pad / 256), blk, 0, c->stream, s->d_cloud.p, s->d_mask.p, n, s->spos.p,
                           s->spos3.p, bnd);
    else
        hipLaunchKernelGGL(k_scene_gather, dim3(n_pad / 256), blk, 0, c->stream, s->d_cloud.p, s->d_tmask.p,
                           s->d_mask.p, s->d_order.p, n, n_pad, s->spos.p, s->spos3.p, s->sref.p, bnd);
    hipLaunchKernelGGL(k_scene_tiles, dim3((n_tiles + 3) / 4), blk, 0, c->stream, s->spos.p, n_tiles, s->tiles.p);
    hipLaunchKernelGGL(k_scene_subs, dim3((n_tiles + 3) / 4), blk, 0, c->stream, s->spos.p, n_tiles, s->subs.p);
    hipLaunchKernelGGL(k_scene_supers, dim3(n_super), blk, 0, c->stream, s->spos.p, n_pad, 4096u, s->supers.p);
    hipLaunchKernelGGL(k_scene_supers, dim3(n_pad / 512), blk, 0, c->stream, s->spos.p, n_pad, 512u, s->chunks.p);
    TM_HIP_CHECK(hipGetLastError());
    if (!resort && s->n_sorted == n_pad) {
        /* only mask_ changed: the arrays are patched in stream order, and the one number the host keeps (the live
         * count, read by the full-scene scoring calls) stays on the device until somebody asks — a search changes the
         * mask twice and waited here for six small kernels each time */
        s->live_pending = true;
        s->n_tiles = n_tiles;
        return TM_OK;
    }
    scene_bounds h;
    TM_HIP_CHECK(hipMemcpyAsync(&h, bnd, sizeof h, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    s->coord_max = h.cmax ? [&] {
        float f;
        memcpy(&f, &h.cmax, 4);
        return f;
    }() : 0.f;
    s->n_live = h.live;
    s->live_pending = false;
    s->n_sorted = n_pad;
    s->n_tiles = n_tiles;
    return TM_OK;
}

int scene_live_count(const tm_scene* cs, uint64_t* live) {
    tm_scene* s = const_cast<tm_scene*>(cs);
    if (s->live_pending) {
        tm_ctx* c = s->ctx;
        TM_HIP_CHECK(hipSetDevice(c->device));
        scene_bounds h;
        TM_HIP_CHECK(hipMemcpyAsync(&h, s->d_bounds.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
        TM_HIP_CHECK(hipStreamSynchronize(c->stream));
        s->n_live = h.live;
        s->live_pending = false;
    }
    *live = s->n_live;
    return TM_OK;
}

}  // namespace tmx

using namespace tmx;

extern "C" int tm_scene_radius_subset(tm_scene* s, uint32_t center_idx, float radius, int32_t* out, uint32_t cap,
                                      uint32_t* count) {
    TM_GUARD(s);
    TM_REQUIRE(s && count, TM_ERR_INVALID, "tm_scene_radius_subset: null argument");
    TM_REQUIRE(center_idx < s->n, TM_ERR_INVALID, "tm_scene_radius_subset: centre index out of range");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t n_words = (s->n + 63) / 64, n_blocks = (s->n + 255) / 256;
    TM_FORWARD(s->w_words.ensure((size_t)n_blocks * 4));
    hipLaunchKernelGGL(k_radius_mask, dim3(n_blocks), dim3(256), 0, c->stream, s->opos.p, s->n, center_idx,
                       radius * radius, s->w_words.p);
    TM_HIP_CHECK(hipGetLastError());
    s->h_words.resize(n_words);
    TM_HIP_CHECK(hipMemcpyAsync(s->h_words.data(), s->w_words.p, (size_t)n_words * 8, hipMemcpyDeviceToHost,
                                c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    uint32_t k = 0;
    if (!out) { /* count only: the list stays described by the membership words (tm_scene_radius_select) */
        for (uint32_t w = 0; w < n_words; ++w) k += (uint32_t)__builtin_popcountll(s->h_words[w]);
        *count = k;
        return TM_OK;
    }
    for (uint32_t w = 0; w < n_words; ++w) {
        unsigned long long b = s->h_words[w];
        while (b) {
            const uint32_t i = w * 64u + (uint32_t)__builtin_ctzll(b);
            if (out && k < cap) out[k] = (int32_t)i;
            ++k;
            b &= b - 1;
        }
    }
    *count = k;
    return TM_OK;
}

/* entries ranks[0..n) (ascending order not required) of the list the LAST tm_scene_radius_subset call on this scene
 * described — for a caller that needs a few members of a large neighbourhood and not the list */
extern "C" int tm_scene_radius_select(const tm_scene* s, const uint32_t* ranks, uint32_t n_ranks, int32_t* out) {
    TM_GUARD(s);
    TM_REQUIRE(s && (ranks || n_ranks == 0) && (out || n_ranks == 0), TM_ERR_INVALID, "tm_scene_radius_select: null argument");
    TM_REQUIRE(!s->h_words.empty(), TM_ERR_NOT_INITIALIZED, "tm_scene_radius_select: no tm_scene_radius_subset call before");
    for (uint32_t q = 0; q < n_ranks; ++q) {
        uint32_t left = ranks[q];
        out[q] = -1;
        for (size_t w = 0; w < s->h_words.size(); ++w) {
            unsigned long long b = s->h_words[w];
            const uint32_t pc = (uint32_t)__builtin_popcountll(b);
            if (left >= pc) {
                left -= pc;
                continue;
            }
            while (left--) b &= b - 1; /* drop the lower set bits */
            out[q] = (int32_t)(w * 64u + (uint32_t)__builtin_ctzll(b));
            break;
        }
        TM_REQUIRE(out[q] >= 0, TM_ERR_INVALID, "tm_scene_radius_select: rank beyond the neighbourhood");
    }
    return TM_OK;
}

namespace tmx {
struct tmask_set {
    __host__ __device__ bool operator()(int32_t v) const { return v != 0; }
};
}  // namespace tmx

int tmx::scene_tangent_index(tm_scene* s) {
    if (s->tangent_idx_ready) return TM_OK;
    tm_ctx* c = s->ctx;
    TM_FORWARD(s->d_tangent_idx.ensure((size_t)s->n + 1));
    size_t tmp = 0;
    rocprim::counting_iterator<uint32_t> idx(0u);
    auto flags = rocprim::make_transform_iterator(s->d_tmask.p, tmask_set());
    TM_HIP_CHECK(rocprim::select(nullptr, tmp, idx, flags, s->d_tangent_idx.p, s->d_tangent_idx.p + s->n, (size_t)s->n,
                                 c->stream));
    TM_FORWARD(scratch_ensure(c, tmp));
    TM_HIP_CHECK(rocprim::select(c->scratch.p, tmp, idx, flags, s->d_tangent_idx.p, s->d_tangent_idx.p + s->n, (size_t)s->n,
                                 c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(&s->n_tangent, s->d_tangent_idx.p + s->n, 4, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    s->tangent_idx_ready = true;
    return TM_OK;
}

int tmx::scene_mirror(const tm_scene* s) {
    if (s->h_sparse || s->h_cloud.size() == (size_t)s->n * 12) return TM_OK;
    TM_HIP_CHECK(hipSetDevice(s->ctx->device));
    s->h_cloud.resize((size_t)s->n * 12);
    hipError_t e = hipMemcpyAsync(s->h_cloud.data(), s->d_cloud.p, (size_t)s->n * 48, hipMemcpyDeviceToHost, s->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->ctx->stream);
    if (e != hipSuccess) {
        s->h_cloud.clear();
        set_error("scene mirror: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

/* the host mirror of mask_, brought up to date from the device copy when tm_scene_mask_add has changed that one */
int tmx::scene_host_mask(const tm_scene* cs) {
    tm_scene* s = const_cast<tm_scene*>(cs);
    if (!s->h_mask_stale) return TM_OK;
    TM_HIP_CHECK(hipSetDevice(s->ctx->device));
    s->h_mask.resize(s->n);
    hipError_t e = hipMemcpyAsync(s->h_mask.data(), s->d_mask.p, (size_t)s->n * 4, hipMemcpyDeviceToHost, s->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->ctx->stream);
    if (e != hipSuccess) {
        set_error("scene mask mirror: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    s->h_mask_stale = false;
    return TM_OK;
}

const float* tmx::scene_row(const tm_scene* s, uint32_t i) {
    if (!s->h_sparse) {
        if (s->h_cloud.empty() && scene_mirror(s) != TM_OK) return nullptr;
        return s->h_cloud.data() + (size_t)i * 12;
    }
    auto it = std::lower_bound(s->h_tan_idx.begin(), s->h_tan_idx.end(), i);
    if (it == s->h_tan_idx.end() || *it != i) return nullptr;
    return s->h_tan_rows.data() + 12 * (size_t)(it - s->h_tan_idx.begin());
}
