// L1-miss / L2-hit gather micro-benchmark for gfx950: cost of a 64-lane gather whose lines come from L2 (a table of
// `MB` MiB, far beyond the 32 KB L1, inside the 4 MB L2 of an XCD or beyond it), by width and by how many lanes share
// a line. The grouped scoring kernel's gathers mostly miss L1 (7e8 L1->L2 requests per launch).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_l2.hip -o tools/_bin/ubench_l2
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

__device__ __forceinline__ uint32_t rng(uint32_t& s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}

// SHARE lanes in a row read the same 128-B line (different dwords of it unless SAME): SHARE = 1 every lane its own line.
// ADJ: the sharing lanes are adjacent (lane / SHARE) or strided (lane % (64 / SHARE))
template <int WIDTH, int SHARE, bool ADJ>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ tab, uint32_t lines_mask, uint32_t iters,
                                                uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t group = ADJ ? lane / SHARE : lane % (64 / SHARE);
    const uint32_t within = ADJ ? lane % SHARE : lane / (64 / SHARE);
    uint32_t seed = (blockIdx.x * 256u + (threadIdx.x & ~63u)) * 2654435761u + 12345u + group * 97u; // same for the sharing lanes
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t line = rng(seed) & lines_mask;
            const uint32_t dw = line * 32u + ((within * 4u) & 31u);
            if (WIDTH == 1) {
                acc += tab[dw];
            } else if (WIDTH == 2) {
                const uint2 v = *reinterpret_cast<const uint2*>(tab + dw);
                acc += v.x + v.y;
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(tab + dw);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int WIDTH, int SHARE, bool ADJ>
static void run(const char* name, const uint32_t* d_tab, uint32_t lines_mask, uint32_t* d_out, int cus) {
    const uint32_t iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_gather<WIDTH, SHARE, ADJ>), dim3(cus * 4), dim3(256), 0, 0, d_tab, lines_mask, iters, d_out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double instr_per_cu = 16.0 * iters * 4.0;
    const double ns = best * 1e6 / instr_per_cu;
    const double lines = 64.0 / SHARE;
    printf("  %-58s %8.3f ms  %7.2f ns per wave gather and CU  %6.2f ns per distinct line  %6.1f GB/s of lines per CU\n", name,
           best, ns, ns / lines, lines * 128.0 / ns);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    uint32_t* d_out;
    CK(hipMalloc(&d_out, 1024));
    for (int mb : {1, 16, 512}) {
        const size_t bytes = (size_t)mb << 20;
        uint32_t* d_tab;
        CK(hipMalloc(&d_tab, bytes + 256));
        CK(hipMemset(d_tab, 0, bytes + 256));
        const uint32_t lines_mask = (uint32_t)(bytes / 128) - 1u;
        printf("-- table of %d MiB (random lines; 16 waves per CU, 4 independent gathers per turn)\n", mb);
        run<1, 1, true>("dword, every lane its own line", d_tab, lines_mask, d_out, cus);
        run<2, 1, true>("dwordx2, every lane its own line", d_tab, lines_mask, d_out, cus);
        run<4, 1, true>("dwordx4, every lane its own line", d_tab, lines_mask, d_out, cus);
        run<1, 4, true>("dword, 4 ADJACENT lanes per line (16 lines)", d_tab, lines_mask, d_out, cus);
        run<1, 4, false>("dword, 4 lanes 16 apart per line (16 lines)", d_tab, lines_mask, d_out, cus);
        run<4, 4, true>("dwordx4, 4 ADJACENT lanes per line (16 lines)", d_tab, lines_mask, d_out, cus);
        run<4, 4, false>("dwordx4, 4 lanes 16 apart per line (16 lines)", d_tab, lines_mask, d_out, cus);
        run<1, 16, true>("dword, 16 adjacent lanes per line (4 lines)", d_tab, lines_mask, d_out, cus);
        run<1, 16, false>("dword, 16 lanes 4 apart per line (4 lines)", d_tab, lines_mask, d_out, cus);
        run<1, 2, true>("dword, 2 adjacent lanes per line (32 lines)", d_tab, lines_mask, d_out, cus);
        CK(hipFree(d_tab));
    }
    return 0;
}
