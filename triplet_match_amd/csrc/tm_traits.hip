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

}  // extern "C"
