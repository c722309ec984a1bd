/* tm_traits.hip — traits projector API (host scalar functions) and the batched cloud projection. */
#include "tm_internal.h"
#include "tm_traits_math.h"

namespace tmx {

/* one thread per point: uvw position, traits tangent, traits normal written as a surfel */
__global__ void k_project_cloud(int kind, const float* __restrict__ state, const float* __restrict__ in,
                                uint32_t n, float* __restrict__ out, int32_t* __restrict__ valid) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    frame_t f = frame_load(state);
    const float* s = in + (size_t)t * 12;
    f3 pos = make3(s[0], s[1], s[2]), nrm = make3(s[4], s[5], s[6]), tgt = make3(s[9], s[10], s[11]);
    f3 uvw;
    bool ok = traits_project(kind, f, pos, uvw);
    float* o = out + (size_t)t * 12;
    if (!ok) {
        for (int k = 0; k < 12; ++k) o[k] = 0.f;
        valid[t] = 0;
        return;
    }
    bool has_t = len(tgt) > 0.f;
    f3 tt = has_t ? traits_tangent(kind, f, pos, tgt) : make3(0.f, 0.f, 0.f);
    /* the cylinder normal() needs a tangent to span the surface; without one keep the mapped normal */
    f3 nn = (kind == 3 && !has_t) ? unit(frame_dir(f, nrm)) : traits_normal(kind, f, pos, nrm, tgt);
    o[0] = uvw.x;
    o[1] = uvw.y;
    o[2] = uvw.z;
    o[3] = 1.f;
    o[4] = nn.x;
    o[5] = nn.y;
    o[6] = nn.z;
    o[7] = 0.f;
    o[8] = s[8];
    o[9] = tt.x;
    o[10] = tt.y;
    o[11] = tt.z;
    valid[t] = 1;
}

/* turns k_project_cloud's output into the inputs of a scene: points whose project() is nullopt get a NaN
 * x (they never take part), tangent_mask = valid && |traits tangent| > 0.7 (the reference's tangent test,
 * include/impl/scene.hpp:49-52, on the projected tangent), mask_ is inherited */
__global__ __launch_bounds__(256) void k_projected_masks(float* __restrict__ cloud, const int32_t* __restrict__ valid,
                                                         const int32_t* __restrict__ src_mask, uint32_t n,
                                                         int32_t* __restrict__ tmask, int32_t* __restrict__ mask) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float* o = cloud + (size_t)i * 12;
    const bool ok = valid[i] != 0;
    if (!ok) o[0] = __builtin_nanf("");
    tmask[i] = (ok && len(make3(o[9], o[10], o[11])) > 0.7f) ? 1 : 0;
    mask[i] = src_mask[i];
}

/* rows (three float4 each) of the listed points, compactly */
__global__ __launch_bounds__(256) void k_gather_rows(const float4* __restrict__ cloud, const uint32_t* __restrict__ idx,
                                                     uint32_t n_idx, float4* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_idx * 3u) return;
    out[t] = cloud[(size_t)idx[t / 3u] * 3u + t % 3u];
}

/* tm_scene_probe_projected: live = finite and not masked (k_scene_gather's rule), tangent = tangent_mask, not
 * masked, within the band of the surface */
__global__ __launch_bounds__(256) void k_probe_count(const float* __restrict__ cloud, const int32_t* __restrict__ tmask,
                                                     const int32_t* __restrict__ mask, uint32_t n, float band,
                                                     uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    bool live = false, tan = false;
    if (i < n) {
        const float* o = cloud + (size_t)i * 12;
        const bool fin = (o[0] - o[0] == 0.f) && (o[1] - o[1] == 0.f) && (o[2] - o[2] == 0.f);
        live = fin && mask[i] == 0;
        tan = tmask[i] != 0 && mask[i] == 0 && (band < 0.f || fabsf(o[2]) <= band);
    }
    const unsigned long long bl = __ballot(live), bt = __ballot(tan);
    if ((threadIdx.x & 63) == 0) {
        if (bl) atomicAdd(&counts[0], (unsigned int)__popcll(bl));
        if (bt) atomicAdd(&counts[1], (unsigned int)__popcll(bt));
    }
}

/* column-major 4x4 * float4, left-to-right sum like opencl/util.cl:4-7 */
__device__ __forceinline__ float4 cl_mat44(float4 p, const float* __restrict__ m) {
    return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * p.w,
                       m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * p.w,
                       m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * p.w,
                       m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15] * p.w);
}

/* opencl/cylinder.cl:1-25 */
__device__ __forceinline__ float4 cl_uv_project(float4 loc, const float* __restrict__ cyl2ncoord) {
    float4 nc = cl_mat44(loc, cyl2ncoord);
    float u = atan2_def(nc.y, nc.x) / 3.14159265358979323846f; /* atan2pi */
    if (u < 0.f) u += 2.f;
    u /= 2.f;
    return make_float4(u, nc.z, sqrtf(nc.x * nc.x + nc.y * nc.y) - 1.0f, 1.f);
}

/* opencl/icp.cl:1-53: one thread per scene point, float4 coalesced load, one image gather */
__global__ void k_icp_projection(const float4* __restrict__ pnts, int n, const float4* __restrict__ image,
                                 int2 img_size, int2 img_margin, const float* __restrict__ mats /* 4 x 16 */,
                                 float max_corr_dist, float4* __restrict__ out_positions,
                                 int* __restrict__ model_indices, int* __restrict__ scene_indices,
                                 unsigned int* __restrict__ n_corr) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    bool corr = false;
    if (index < n) {
        const float* mat_align = mats;
        const float* mat_uvw = mats + 16;
        const float* mat_proj = mats + 32;
        const float* mat_norm = mats + 48;
        float4 loc = cl_mat44(pnts[index], mat_align);
        float4 uv = cl_uv_project(loc, mat_proj);
        float4 uvn = cl_mat44(cl_mat44(uv, mat_norm), mat_uvw);
        float extx = (float)(img_size.x - 2 * img_margin.x - 1), exty = (float)(img_size.y - 2 * img_margin.y - 1);
        int px = (int)(uvn.x * extx) + img_margin.x;
        int py = (int)(uvn.y * exty) + img_margin.y;
        if (py == img_size.y) py = img_size.y - 1;
        int mi = -1, si = -1;
        float4 outp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (px >= 0 && px < img_size.x && py >= 0 && py < img_size.y) {
            int idx = py * img_size.x + px;
            float4 im = image[idx];
            float dx = im.x - uvn.x, dy = im.y - uvn.y;
            float dist = sqrtf(dx * dx + dy * dy);
            outp.w = dist;
            if (dist < max_corr_dist) {
                mi = idx;
                si = index;
                outp = uvn;
                corr = true;
            }
        }
        model_indices[index] = mi;
        scene_indices[index] = si;
        out_positions[index] = outp;
    }
    unsigned long long b = __ballot(corr);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_corr, (unsigned int)__popcll(b));
}

/* opencl/icp.cl:55-86 fused with its missing reduction: per-thread products in f64, wave shuffle,
 * one atomic per wave and matrix entry */
__global__ void k_icp_correlation(const float4* __restrict__ scene, const float4* __restrict__ model,
                                  const int* __restrict__ is, const int* __restrict__ im, int n, float4 cs,
                                  float4 cm, double* __restrict__ cov9) {
    const int index = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    if (index < n) {
        float4 s = scene[is[index]], m = model[im[index]];
        float sx = s.x - cs.x, sy = s.y - cs.y, sz = s.z - cs.z;
        float mx = m.x - cm.x, my = m.y - cm.y, mz = m.z - cm.z;
        float norm = 1.f / (float)(n - 1);
        acc[0] = sx * mx * norm; acc[1] = sy * mx * norm; acc[2] = sz * mx * norm;
        acc[3] = sx * my * norm; acc[4] = sy * my * norm; acc[5] = sz * my * norm;
        acc[6] = sx * mz * norm; acc[7] = sy * mz * norm; acc[8] = sz * mz * norm;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off);
        if ((threadIdx.x & 63) == 0 && acc[k] != 0.0) atomicAdd(&cov9[k], acc[k]);
    }
}

}  // namespace tmx

using namespace tmx;

extern "C" {

int tm_traits_init_from_samples(int kind, const float* model_state, const float* samples, float* state) {
    if (kind < 0 || kind > 3 || !model_state || !samples || !state) return 1;
    return traits_init_from_samples(kind, model_state, samples, state) ? 0 : 1;
}

int tm_traits_project(int kind, const float* state, const float xyz[3], float uvw[3]) {
    frame_t f = frame_load(state);
    f3 o;
    if (!traits_project(kind, f, make3(xyz[0], xyz[1], xyz[2]), o)) return 0;
    uvw[0] = o.x;
    uvw[1] = o.y;
    uvw[2] = o.z;
    return 1;
}

/* reference: identity_traits.hpp:38-41, plane_traits.hpp:75-78, plane2_traits.hpp:90-93,
 * cylinder_traits.hpp:117-131 */
void tm_traits_unproject(int kind, const float* state, const float uvw[3], float xyz[3]) {
    if (kind == TM_TRAITS_IDENTITY) {
        xyz[0] = uvw[0];
        xyz[1] = uvw[1];
        xyz[2] = uvw[2];
        return;
    }
    float lx = uvw[0], ly = uvw[1], lz = uvw[2];
    if (kind == TM_TRAITS_CYLINDER) {
        float radius = state[0];
        float r = (uvw[2] + 1.f) * radius;
        float ang = uvw[0] / radius;
        lx = r * cosf(ang);
        ly = r * sinf(ang);
        lz = uvw[1];
    }
    const float* l = state + 21;
    for (int r = 0; r < 3; ++r) xyz[r] = ((l[r] * lx + l[4 + r] * ly) + l[8 + r] * lz) + l[12 + r];
}

void tm_traits_tangent(int kind, const float* state, const float* s, float out[3]) {
    frame_t f = frame_load(state);
    f3 r = traits_tangent(kind, f, make3(s[0], s[1], s[2]), make3(s[9], s[10], s[11]));
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}

void tm_traits_normal(int kind, const float* state, const float* s, float out[3]) {
    frame_t f = frame_load(state);
    f3 r = traits_normal(kind, f, make3(s[0], s[1], s[2]), make3(s[4], s[5], s[6]),
                         make3(s[9], s[10], s[11]));
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}

float tm_traits_intrinsic_distance(int kind, const float* state, const float p0[3], const float p1[3]) {
    return traits_distance(kind, state[0], make3(p0[0], p0[1], p0[2]), make3(p1[0], p1[1], p1[2]));
}

int tm_traits_project_cloud(tm_ctx* c, int kind, const float* state, const float* surfels, uint32_t n,
                            float* uvw_surfels, int32_t* valid) {
    TM_GUARD(c);
    TM_REQUIRE(c && state && surfels && uvw_surfels && valid, TM_ERR_INVALID,
               "tm_traits_project_cloud: null argument");
    TM_REQUIRE(kind >= 0 && kind <= 3, TM_ERR_INVALID, "bad traits kind");
    if (n == 0) return TM_OK;
    TM_HIP_CHECK(hipSetDevice(c->device));
    dbuf<float> d_state, d_in, d_out;
    dbuf<int32_t> d_valid;
    int rc;
    auto cleanup = [&]() {
        d_state.release();
        d_in.release();
        d_out.release();
        d_valid.release();
    };
    if ((rc = d_state.alloc(TM_STATE_FLOATS)) || (rc = d_in.alloc((size_t)n * 12)) ||
        (rc = d_out.alloc((size_t)n * 12)) || (rc = d_valid.alloc(n))) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d_state.p, state, TM_STATE_FLOATS * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_in.p, surfels, (size_t)n * 48, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_project_cloud, dim3((n + 255) / 256), dim3(256), 0, c->stream, kind, d_state.p,
                           d_in.p, n, d_out.p, d_valid.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(uvw_surfels, d_out.p, (size_t)n * 48, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(valid, d_valid.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) {
        set_error("tm_traits_project_cloud: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

int tm_scene_create_projected(tm_scene* src, int kind, const float* state, tm_scene** out) {
    TM_GUARD(src);
    TM_REQUIRE(src && state && out, TM_ERR_INVALID, "tm_scene_create_projected: null argument");
    TM_REQUIRE(kind >= 0 && kind <= 3, TM_ERR_INVALID, "bad traits kind");
    tm_ctx* c = src->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t n = src->n;
    tm_scene* t = new tm_scene();
    t->ctx = c;
    t->n = n;
    dbuf<float> d_state;
    dbuf<int32_t> d_valid;
    int rc = TM_OK;
    hipError_t e = hipSuccess;
    if ((rc = d_state.alloc(TM_STATE_FLOATS)) || (rc = d_valid.alloc(n)) || (rc = t->d_cloud.ensure((size_t)n * 12)) ||
        (rc = t->d_tmask.ensure(n)) || (rc = t->d_mask.ensure(n))) {
    } else {
        e = hipMemcpyAsync(d_state.p, state, TM_STATE_FLOATS * 4, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_project_cloud, dim3((n + 255) / 256), dim3(256), 0, c->stream, kind, d_state.p,
                               src->d_cloud.p, n, t->d_cloud.p, d_valid.p);
            hipLaunchKernelGGL(k_projected_masks, dim3((n + 255) / 256), dim3(256), 0, c->stream, t->d_cloud.p,
                               d_valid.p, src->d_mask.p, n, t->d_tmask.p, t->d_mask.p);
            e = hipGetLastError();
        }
        /* Host mirrors (sample filters, subset bookkeeping). The host side of a search reads the surfels of
         * tangent points only, so only those travel: the ascending tangent index list (kept on the device for the
         * neighbourhood views as well), the gathered rows, and tangent_mask rebuilt from the list — 1.7 MB instead of
         * 52 MB per projector at a million points with 4 % tangent points. */
        if (tmx::scene_host_mask(src) != TM_OK) e = hipErrorUnknown;
        t->h_mask = src->h_mask;
        t->h_sparse = true;
        if (e != hipSuccess) {
            set_error("tm_scene_create_projected: %s", hipGetErrorString(e));
            rc = TM_ERR_HIP;
        }
        if (!rc) rc = scene_tangent_index(t);
        if (!rc && t->n_tangent) {
            dbuf<float> d_rows;
            rc = d_rows.alloc((size_t)t->n_tangent * 12);
            if (!rc) {
                hipLaunchKernelGGL(k_gather_rows, dim3((t->n_tangent * 3u + 255u) / 256u), dim3(256), 0, c->stream,
                                   (const float4*)t->d_cloud.p, t->d_tangent_idx.p, t->n_tangent, (float4*)d_rows.p);
                t->h_tan_idx.resize(t->n_tangent);
                t->h_tan_rows.resize((size_t)t->n_tangent * 12);
                e = hipGetLastError();
                if (e == hipSuccess)
                    e = hipMemcpyAsync(t->h_tan_idx.data(), t->d_tangent_idx.p, (size_t)t->n_tangent * 4, hipMemcpyDeviceToHost,
                                       c->stream);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(t->h_tan_rows.data(), d_rows.p, (size_t)t->n_tangent * 48, hipMemcpyDeviceToHost,
                                       c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) {
                    set_error("tm_scene_create_projected: %s", hipGetErrorString(e));
                    rc = TM_ERR_HIP;
                }
            }
            d_rows.release();
        }
        if (!rc) {
            t->h_tmask.assign(n, 0);
            for (uint32_t i : t->h_tan_idx) t->h_tmask[i] = 1;
        }
    }
    d_state.release();
    d_valid.release();
    if (!rc) rc = scene_build(t, true);
    if (rc) {
        tm_scene_destroy(t);
        return rc;
    }
    *out = t;
    return TM_OK;
}

/* What tm_scene_create_projected would give, counted without building it: the points that project and are not
 * masked (tm_scene_live_count of the projected scene) and the tangent points among them within `band` of the
 * surface (abs(w) <= band; band < 0 = no test). The same two kernels into scratch memory + a count. */
int tm_scene_probe_projected(tm_scene* src, int kind, const float* state, float band, uint64_t* live,
                             uint32_t* tangent_on_surface) {
    TM_GUARD(src);
    TM_REQUIRE(src && state && live && tangent_on_surface, TM_ERR_INVALID, "tm_scene_probe_projected: null argument");
    TM_REQUIRE(kind >= 0 && kind <= 3, TM_ERR_INVALID, "bad traits kind");
    tm_ctx* c = src->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t n = src->n;
    *live = 0;
    *tangent_on_surface = 0;
    if (n == 0) return TM_OK;
    /* scratch: projected cloud (12 floats per point), valid, tmask, mask, state, two counters */
    const size_t bytes = (size_t)n * 48 + (size_t)n * 12 + TM_STATE_FLOATS * 4 + 16;
    TM_FORWARD(scratch_ensure(c, bytes));
    float* d_cloud = reinterpret_cast<float*>(c->scratch.p);
    int32_t* d_valid = reinterpret_cast<int32_t*>(d_cloud + (size_t)n * 12);
    int32_t* d_tmask = d_valid + n;
    int32_t* d_mask = d_tmask + n;
    float* d_state = reinterpret_cast<float*>(d_mask + n);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(d_state + TM_STATE_FLOATS);
    TM_HIP_CHECK(hipMemcpyAsync(d_state, state, TM_STATE_FLOATS * 4, hipMemcpyHostToDevice, c->stream));
    TM_HIP_CHECK(hipMemsetAsync(d_cnt, 0, 8, c->stream));
    hipLaunchKernelGGL(k_project_cloud, dim3((n + 255) / 256), dim3(256), 0, c->stream, kind, d_state, src->d_cloud.p, n,
                       d_cloud, d_valid);
    hipLaunchKernelGGL(k_projected_masks, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_cloud, d_valid, src->d_mask.p,
                       n, d_tmask, d_mask);
    hipLaunchKernelGGL(k_probe_count, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_cloud, d_tmask, d_mask, n, band,
                       d_cnt);
    TM_HIP_CHECK(hipGetLastError());
    uint32_t h[2] = {0, 0};
    TM_HIP_CHECK(hipMemcpyAsync(h, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    *live = h[0];
    *tangent_on_surface = h[1];
    return TM_OK;
}

int tm_cl_icp_projection(tm_ctx* c, const float* pnts4, int n, const float* image4, const int img_size[2],
                         const int img_margin[2], const float* mat_align16, const float* mat_uvw16,
                         const float* mat_proj16, const float* mat_norm16, float max_corr_dist,
                         float* out_positions4, int* model_indices, int* scene_indices, uint32_t* n_corr) {
    TM_GUARD(c);
    TM_REQUIRE(c && pnts4 && image4 && img_size && img_margin && mat_align16 && mat_uvw16 && mat_proj16 &&
                   mat_norm16 && out_positions4 && model_indices && scene_indices,
               TM_ERR_INVALID, "tm_cl_icp_projection: null argument");
    TM_REQUIRE(n >= 0 && img_size[0] > 0 && img_size[1] > 0, TM_ERR_INVALID, "tm_cl_icp_projection: bad sizes");
    if (n_corr) *n_corr = 0;
    if (n == 0) return TM_OK;
    TM_HIP_CHECK(hipSetDevice(c->device));
    const size_t npix = (size_t)img_size[0] * img_size[1];
    dbuf<float4> d_p, d_img, d_out;
    dbuf<float> d_m;
    dbuf<int> d_mi, d_si;
    dbuf<unsigned int> d_n;
    auto cleanup = [&]() {
        d_p.release(); d_img.release(); d_out.release(); d_m.release(); d_mi.release(); d_si.release(); d_n.release();
    };
    int rc;
    if ((rc = d_p.alloc(n)) || (rc = d_img.alloc(npix)) || (rc = d_out.alloc(n)) || (rc = d_m.alloc(64)) ||
        (rc = d_mi.alloc(n)) || (rc = d_si.alloc(n)) || (rc = d_n.alloc(1))) {
        cleanup();
        return rc;
    }
    float mats[64];
    memcpy(mats, mat_align16, 64);
    memcpy(mats + 16, mat_uvw16, 64);
    memcpy(mats + 32, mat_proj16, 64);
    memcpy(mats + 48, mat_norm16, 64);
    hipError_t e = hipMemcpyAsync(d_p.p, pnts4, (size_t)n * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_img.p, image4, npix * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_m.p, mats, sizeof(mats), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_n.p, 0, 4, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_icp_projection, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_p.p, n, d_img.p,
                           make_int2(img_size[0], img_size[1]), make_int2(img_margin[0], img_margin[1]), d_m.p,
                           max_corr_dist, d_out.p, d_mi.p, d_si.p, d_n.p);
        e = hipGetLastError();
    }
    unsigned int h_n = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(out_positions4, d_out.p, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(model_indices, d_mi.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(scene_indices, d_si.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_n, d_n.p, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) {
        set_error("tm_cl_icp_projection: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    if (n_corr) *n_corr = h_n;
    return TM_OK;
}

int tm_cl_icp_correlation(tm_ctx* c, const float* scene4, uint32_t n_scene, const float* model4, uint32_t n_model,
                          const int* indices_scene, const int* indices_model, int n, const float centroid_scene[4],
                          const float centroid_model[4], double cov9[9]) {
    TM_GUARD(c);
    TM_REQUIRE(c && scene4 && model4 && indices_scene && indices_model && centroid_scene && centroid_model && cov9,
               TM_ERR_INVALID, "tm_cl_icp_correlation: null argument");
    TM_REQUIRE(n >= 2, TM_ERR_INVALID, "tm_cl_icp_correlation: needs at least two correspondences");
    for (int i = 0; i < n; ++i)
        TM_REQUIRE(indices_scene[i] >= 0 && (uint32_t)indices_scene[i] < n_scene && indices_model[i] >= 0 &&
                       (uint32_t)indices_model[i] < n_model,
                   TM_ERR_INVALID, "tm_cl_icp_correlation: index out of range");
    TM_HIP_CHECK(hipSetDevice(c->device));
    dbuf<float4> d_s, d_m;
    dbuf<int> d_is, d_im;
    dbuf<double> d_cov;
    auto cleanup = [&]() { d_s.release(); d_m.release(); d_is.release(); d_im.release(); d_cov.release(); };
    int rc;
    if ((rc = d_s.alloc(n_scene)) || (rc = d_m.alloc(n_model)) || (rc = d_is.alloc(n)) || (rc = d_im.alloc(n)) ||
        (rc = d_cov.alloc(9))) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d_s.p, scene4, (size_t)n_scene * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_m.p, model4, (size_t)n_model * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_is.p, indices_scene, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_im.p, indices_model, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_cov.p, 0, 72, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_icp_correlation, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_s.p, d_m.p, d_is.p,
                           d_im.p, n,
                           make_float4(centroid_scene[0], centroid_scene[1], centroid_scene[2], centroid_scene[3]),
                           make_float4(centroid_model[0], centroid_model[1], centroid_model[2], centroid_model[3]),
                           d_cov.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(cov9, d_cov.p, 72, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) {
        set_error("tm_cl_icp_correlation: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

}  // extern "C"
