// VALU / LDS issue-cost micro-benchmark for gfx950: cycles per wave-instruction of the instruction kinds the grouped
// scoring kernel is made of, with 1, 2 and 4 waves per SIMD, so that its per-stage instruction mix (tools/isa_mix.py)
// can be priced in SIMD cycles instead of counted in instructions.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                               \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

// 8 independent accumulators so that the issue rate, not a dependency chain, is what is measured
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(uint32_t iters, float* out, unsigned long long* clk) {
    __shared__ float4 s_pose[64];
    __shared__ unsigned long long s_acc[16];
    if (threadIdx.x < 64) s_pose[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
    if (threadIdx.x < 16) s_acc[threadIdx.x] = 0ull;
    __syncthreads();
    float a[8], b = 1.0001f + 1e-9f * threadIdx.x, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
    uint32_t u[8];
    unsigned long long w[8];
    double d[8];
    for (int k = 0; k < 8; ++k) {
        a[k] = 1.f + k + 1e-3f * threadIdx.x;
        p[k] = f2{a[k], a[k] + 1.f};
        u[k] = threadIdx.x * 7u + k;
        w[k] = (unsigned long long)u[k] << 20;
        d[k] = (double)a[k];
    }
    const f2 pb = f2{b, b};
    /* s_memtime (shader clock) against s_memrealtime (100 MHz): the clock this kernel really ran at */
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (uint32_t it = 0; it < iters; ++it) {
        if (KIND == 0) {
#define OP(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            REP8(OP)
#undef OP
        } else if (KIND == 1) {
#define OP(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
            REP8(OP)
#undef OP
        } else if (KIND == 2) {
#define OP(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
            REP8(OP)
#undef OP
        } else if (KIND == 3) {
#define OP(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
            REP8(OP)
#undef OP
        } else if (KIND == 4) {
#define OP(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[k]) : "v"(u[k]), "v"(u[(k + 1) & 7]) : "vcc");
            REP8(OP)
#undef OP
        } else if (KIND == 5) {
#define OP(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "v"(u[(k + 2) & 7]));
            REP8(OP)
#undef OP
        } else if (KIND == 6) {
#define OP(k) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[k]) : "v"(a[k]));
            REP8(OP)
#undef OP
        } else if (KIND == 7) {
#define OP(k) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(w[k]) : "v"(u[k] & 3u));
            REP8(OP)
#undef OP
        } else if (KIND == 8) {
#define OP(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
            REP8(OP)
#undef OP
        } else if (KIND == 9) {
#define OP(k) asm volatile("v_trunc_f64 %0, %0" : "+v"(d[k]));
            REP8(OP)
#undef OP
        } else if (KIND == 10) {
#define OP(k) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(u[k]));
            REP8(OP)
#undef OP
        } else if (KIND == 11) { /* LDS: 16-byte read, four distinct addresses per wave (a pose per 16 lanes) */
            const float4* q = s_pose + (threadIdx.x >> 4 & 3) * 3;
#define OP(k)                                                   \
    {                                                           \
        float4 v = q[k & 1];                                    \
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); \
        a[k] += v.x;                                            \
    }
            REP8(OP)
#undef OP
        } else if (KIND == 12) { /* LDS: one 64-bit atomic add per lane, four distinct addresses per wave */
#define OP(k) atomicAdd(&s_acc[(threadIdx.x >> 4) & 3], w[k] | 1ull);
            REP8(OP)
#undef OP
        } else if (KIND == 13) {
#define OP(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a[k]), "v"(b) : "vcc");
            REP8(OP)
#undef OP
        } else if (KIND == 14) {
#define OP(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 7]) : "vcc");
            REP8(OP)
#undef OP
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0, clk[1] = r1 - r0;
    float r = c;
    for (int k = 0; k < 8; ++k) r += a[k] + p[k].x + p[k].y + (float)u[k] + (float)w[k] + (float)d[k];
    if (r == 0.12345f) out[threadIdx.x] = r + (float)s_acc[0];
}

template <int KIND>
static void run(const char* name, int waves_per_simd) {
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float* d_out;
    CK(hipMalloc(&d_out, 1024));
    const uint32_t iters = 20000;
    // one workgroup of 256 threads = 4 waves = one wave per SIMD of a CU; waves_per_simd workgroups per CU
    const dim3 grid((unsigned)(cus * waves_per_simd));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    unsigned long long* d_clk;
    CK(hipMalloc(&d_clk, 16));
    hipLaunchKernelGGL(k_issue<KIND>, grid, dim3(256), 0, 0, 100u, d_out, (unsigned long long*)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_issue<KIND>, grid, dim3(256), 0, 0, iters, d_out, d_clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    // SIMD cycles per wave-instruction = time * clock / (instructions per SIMD) ; instructions per SIMD = iters * 8 * waves
    const double cyc = (double)ms * 1e-3 * (double)khz * 1e3 / ((double)iters * 8.0 * waves_per_simd);
    unsigned long long h_clk[2] = {0, 0};
    CK(hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost));
    /* one wave's view: shader-clock ticks per instruction of all waves of its SIMD, and the shader clock itself */
    const double own = (double)h_clk[0] / ((double)iters * 8.0 * waves_per_simd);
    const double mhz = h_clk[1] ? (double)h_clk[0] / (double)h_clk[1] * 100.0 : 0.0;
    printf("%-44s %d wave(s)/SIMD  %8.3f ms  %6.2f cycles of the nominal %d MHz clock per wave-instruction and SIMD; s_memtime: %.2f ticks, %.0f MHz\n", name,
           waves_per_simd, ms, cyc, khz / 1000, own, mhz);
    CK(hipFree(d_out));
    CK(hipFree(d_clk));
}

#define ALL(K, NAME) \
    run<K>(NAME, 1); \
    run<K>(NAME, 2); \
    run<K>(NAME, 4);

int main() {
    ALL(0, "v_add_f32")
    ALL(1, "v_mul_f32")
    ALL(2, "v_pk_mul_f32")
    ALL(3, "v_pk_add_f32")
    ALL(4, "v_mad_u64_u32")
    ALL(5, "v_mad_u32_u24")
    ALL(6, "v_cvt_i32_f32")
    ALL(7, "v_lshrrev_b64")
    ALL(8, "v_cvt_f64_f32")
    ALL(9, "v_trunc_f64")
    ALL(10, "v_add_u32_dpp row_shr")
    ALL(11, "ds_read_b128, 4 addresses per wave")
    ALL(12, "ds_add_u64, 4 addresses per wave")
    ALL(13, "v_cmp_lt_f32 -> vcc")
    ALL(14, "v_cndmask_b32")
    return 0;
}
