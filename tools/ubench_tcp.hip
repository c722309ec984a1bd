// L1 (TCP) tag-lookup micro-benchmark for gfx950: how does the address pattern of a 64-lane gather change its cost
// when every line is an L1 hit? The grouped scoring kernel runs at ~0.9 of the L1's tag-lookup rate
// (TCP_TOTAL_CACHE_ACCESSES / CU-cycles), so what matters is which lanes the texture addresser merges into one lookup.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_tcp.hip -o /tmp/ubench_tcp ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

// every lane keeps its dword index; UNROLL independent gathers per turn, the index is nudged inside an 8 KB window so
// the loads cannot be hoisted; everything stays L1-resident (window of 2048 dwords per workgroup slot)
template <int WIDTH>  // dwords per lane and gather: 1, 3 (dwordx3) or 4 (dwordx4)
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ tab, const uint32_t* __restrict__ pattern,
                                                uint32_t iters, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t idx = pattern[lane];  // dword index of this lane inside a 2048-dword window; bit 31: lane masked off
    uint32_t acc = 0;
    const bool off = (idx >> 31) != 0u;
    idx &= 0x7FFFFFFFu;
    const uint32_t* base = tab + (size_t)(blockIdx.x & 7u) * 2048u;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t j = (idx + (uint32_t)u * 256u * 0u) & 2047u;  // same lines every time: pure hit rate
            if (off) continue;  // EXEC-masked lanes: what does a partially filled gather cost?
            if (WIDTH == 0) {
                acc += reinterpret_cast<const unsigned char*>(base)[j * 4u];
            } else if (WIDTH == 2) {
                const uint2 v = *reinterpret_cast<const uint2*>(base + (j & ~1u));
                acc += v.x + v.y;
            } else if (WIDTH == 1) {
                acc += base[j];
            } else if (WIDTH == 3) {
                struct f3 { uint32_t a, b, c; };
                const f3 v = *reinterpret_cast<const f3*>(base + (j / 3u) * 3u);
                acc += v.a + v.b + v.c;
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(base + (j & ~3u));
                acc += v.x + v.y + v.z + v.w;
            }
            asm volatile("" : "+v"(idx));  // keep the loads in the loop
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

struct pat {
    const char* name;
    std::vector<uint32_t> idx;
};

int main() {
    std::vector<pat> pats;
    auto add = [&](const char* n, auto f) {
        pat p;
        p.name = n;
        for (uint32_t l = 0; l < 64; ++l) p.idx.push_back(f(l));
        pats.push_back(p);
    };
    add("coalesced: lane -> consecutive dwords (4 segments of 64 B)", [](uint32_t l) { return l; });
    add("64 distinct 64-B segments (stride 16 dwords)", [](uint32_t l) { return l * 16u; });
    add("16 segments, copies 16 lanes apart, SAME dword (l & 15)", [](uint32_t l) { return (l & 15u) * 16u; });
    add("16 segments, copies in ADJACENT lanes, SAME dword (l >> 2)", [](uint32_t l) { return (l >> 2) * 16u; });
    add("16 segments, 16 lanes apart, different dwords of the segment", [](uint32_t l) { return (l & 15u) * 16u + (l >> 4); });
    add("16 segments, adjacent lanes, different dwords of the segment", [](uint32_t l) { return (l >> 2) * 16u + (l & 3u); });
    add("16 segments, adjacent lanes, dwords 4 apart in the segment", [](uint32_t l) { return (l >> 2) * 16u + (l & 3u) * 4u; });
    add("32 segments, pairs of adjacent lanes", [](uint32_t l) { return (l >> 1) * 16u; });
    add("32 segments, pairs 32 lanes apart", [](uint32_t l) { return (l & 31u) * 16u; });
    add("8 segments, 8 adjacent lanes each, same dword", [](uint32_t l) { return (l >> 3) * 16u; });
    add("all lanes one dword", [](uint32_t) { return 0u; });
    add("27 of 64 lanes random segments (others lane 0's)", [](uint32_t l) { return (l % 7u < 3u) ? ((l * 37u) & 127u) * 16u : 0u; });
    const uint32_t OFF = 0x80000000u;
    add("27 lanes random segments, the others MASKED OFF", [OFF](uint32_t l) { return (l % 7u < 3u) ? ((l * 37u) & 127u) * 16u : OFF; });
    add("32 lanes (whole quads 0,2,4..) distinct segments, others masked", [OFF](uint32_t l) { return ((l >> 2) & 1u) ? OFF : l * 16u; });
    add("32 lanes (even lanes) distinct segments, others masked", [OFF](uint32_t l) { return (l & 1u) ? OFF : l * 16u; });
    add("16 lanes (lane 0 of each quad) distinct segments, others masked", [OFF](uint32_t l) { return (l & 3u) ? OFF : l * 16u; });
    add("quads: same dword, every quad its own 128-B line", [](uint32_t l) { return (l >> 2) * 32u; });
    add("quads: 2 distinct dwords of one aligned 16 B, own line each", [](uint32_t l) { return (l >> 2) * 32u + ((l >> 1) & 1u); });
    add("quads: 3 lanes same dword, 1 lane in another line", [](uint32_t l) { return (l & 3u) == 3u ? (l >> 2) * 32u + 1024u : (l >> 2) * 32u; });
    add("quads: 2 + 2 lanes in two lines", [](uint32_t l) { return (l & 2u) ? (l >> 2) * 32u + 1024u : (l >> 2) * 32u; });
    add("quads: same 64-B segment, 4 different 16-B pieces, own line each", [](uint32_t l) { return (l >> 2) * 32u + (l & 3u) * 4u; });
    add("octets (8 adjacent lanes): same dword, own line each", [](uint32_t l) { return (l >> 3) * 32u; });
    add("pairs: same dword, own line each", [](uint32_t l) { return (l >> 1) * 32u; });

    uint32_t *d_tab, *d_pat, *d_out;
    CK(hipMalloc(&d_tab, 8 * 2048 * 4 + 64));
    CK(hipMemset(d_tab, 0, 8 * 2048 * 4 + 64));
    CK(hipMalloc(&d_pat, 64 * 4));
    CK(hipMalloc(&d_out, 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    printf("%d CUs, nominal clock %.2f GHz; 4 workgroups x 256 threads per CU, 8 gathers per turn\n", cus, khz * 1e-6);
    const uint32_t iters = 8000;
    const dim3 grid(cus * 4), block(256);
    for (int width : {0, 1, 2, 3, 4}) {
        printf("-- %d dword(s) per lane (0 = one byte)\n", width);
        for (const pat& p : pats) {
            CK(hipMemcpy(d_pat, p.idx.data(), 64 * 4, hipMemcpyHostToDevice));
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (width == 0) hipLaunchKernelGGL(k_gather<0>, grid, block, 0, 0, d_tab, d_pat, iters, d_out);
                if (width == 2) hipLaunchKernelGGL(k_gather<2>, grid, block, 0, 0, d_tab, d_pat, iters, d_out);
                if (width == 1) hipLaunchKernelGGL(k_gather<1>, grid, block, 0, 0, d_tab, d_pat, iters, d_out);
                if (width == 3) hipLaunchKernelGGL(k_gather<3>, grid, block, 0, 0, d_tab, d_pat, iters, d_out);
                if (width == 4) hipLaunchKernelGGL(k_gather<4>, grid, block, 0, 0, d_tab, d_pat, iters, d_out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            // wave-level gather instructions per CU: 16 waves x iters x 8
            const double instr_per_cu = 16.0 * iters * 8.0;
            const double ns_per_instr = best * 1e6 / instr_per_cu;
            printf("  %-66s %8.3f ms  %6.2f ns per wave gather and CU (= %5.1f cycles at 2.4 GHz)\n", p.name, best,
                   ns_per_instr, ns_per_instr * 2.4);
        }
    }
    return 0;
}
