/*
 * tm_comm.hip — the multi-GPU step of the path: one process per GPU, scene / model /
 * table replicated, hypotheses (or the first-point chunks of find_parallel) sharded over the ranks, and ONE
 * small collective per batch: an RCCL all-gather of 96-byte best records over xGMI followed by the
 * deterministic selection (count desc, score desc, hypothesis id asc). The reference has no counterpart
 * (its only reduction is fut.get() + std::sort over threads, include/impl/scene.hpp:168-195).
 *
 * RCCL is bound at run time (dlopen of librccl.so.1, RTLD_LOCAL): libtm_hip.so carries no link-time
 * dependency on it, and a process that already holds an RCCL (PyTorch does) shares that copy. tm_comm_* return
 * TM_ERR_UNSUPPORTED with a message when no RCCL can be loaded; nothing falls back to another transport.
 */
#include "tm_internal.h"

#include <dlfcn.h>

#include <mutex>

namespace tmx {

/* the few RCCL entry points used, with the types of rccl.h restated (ncclUniqueId = 128 bytes,
 * ncclUint8 = 1, ncclResult_t 0 = success) */
struct rccl_unique_id {
    char internal[128];
};
typedef void* rccl_comm_t;
struct rccl_api {
    void* lib = nullptr;
    int (*GetUniqueId)(rccl_unique_id*) = nullptr;
    int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

static rccl_api& rccl() {
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        /* an RCCL the process already holds (PyTorch's, when torch was imported first) is shared; otherwise
         * load one privately: RTLD_LOCAL, so that its symbols never interpose on a library loaded later (a
         * torch imported after an RTLD_GLOBAL RCCL ended in a double free at process exit) */
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (api.lib) break;
        }
        for (const char* n : names) {
            if (api.lib) break;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!api.lib) return;
        api.GetUniqueId = (int (*)(rccl_unique_id*))dlsym(api.lib, "ncclGetUniqueId");
        api.CommInitRank = (int (*)(rccl_comm_t*, int, rccl_unique_id, int))dlsym(api.lib, "ncclCommInitRank");
        api.CommDestroy = (int (*)(rccl_comm_t))dlsym(api.lib, "ncclCommDestroy");
        api.AllGather =
            (int (*)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t))dlsym(api.lib, "ncclAllGather");
        api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather;
    });
    return api;
}

#define TM_RCCL_CHECK(expr)                                                                           \
    do {                                                                                              \
        int _r = (expr);                                                                              \
        if (_r != 0) {                                                                                \
            tmx::set_error("%s failed: %s", #expr,                                                     \
                          tmx::rccl().GetErrorString ? tmx::rccl().GetErrorString(_r) : "RCCL error"); \
            return TM_ERR_HIP;                                                                        \
        }                                                                                             \
    } while (0)

/* better(a, b): a precedes b in the order count desc, score desc, id asc; records without a hypothesis last */
__host__ __device__ inline bool record_better(const tm_best_record& a, const tm_best_record& b) {
    if (a.hyp_id == 0xFFFFFFFFu) return false;
    if (b.hyp_id == 0xFFFFFFFFu) return true;
    if (a.inlier_count != b.inlier_count) return a.inlier_count > b.inlier_count;
    if (a.score != b.score) return a.score > b.score;
    return a.hyp_id < b.hyp_id;
}

/* all[r * n_models + m] = rank r's record for model m; out[m] = the best of them. One thread per model. */
__global__ void k_best_select(const tm_best_record* __restrict__ all, uint32_t world, uint32_t n_models,
                              tm_best_record* __restrict__ out) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_models) return;
    tm_best_record b = all[m];
    for (uint32_t r = 1; r < world; ++r) {
        const tm_best_record c = all[(size_t)r * n_models + m];
        if (record_better(c, b)) b = c;
    }
    if (b.hyp_id == 0xFFFFFFFFu) { /* canonical empty record */
        b.inlier_count = 0;
        b.score = 0.0;
        for (int k = 0; k < 16; ++k) b.pose[k] = 0.f;
    }
    out[m] = b;
}

/* argmax over a scored batch with ids that do not depend on how the sample pairs were dealt to the ranks:
 * id = global pair index * query_limit + entry within the pair (prov = (local pair, m_i, m_j, entry)).
 * One workgroup; the record stays on the device. */
__global__ __launch_bounds__(1024) void k_best_prov(const uint32_t* __restrict__ counts,
                                                    const double* __restrict__ scores,
                                                    const float* __restrict__ poses, const uint4* __restrict__ prov,
                                                    uint32_t n, uint32_t pair_offset, uint32_t pair_stride,
                                                    uint32_t limit, tm_best_record* __restrict__ out) {
    __shared__ tm_best_record s_rec[16];
    tm_best_record b;
    b.inlier_count = 0;
    b.hyp_id = 0xFFFFFFFFu;
    b.score = 0.0;
    uint32_t bi = 0xFFFFFFFFu;
    /* one workgroup walks the batch: four records per turn, their loads in flight together (64 dependent turns of
     * three loads each were 53 us for 65,536 hypotheses) */
    for (uint32_t i0 = threadIdx.x; i0 < n; i0 += 4096) {
        uint4 pv[4];
        uint32_t cn[4];
        double sc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + (uint32_t)k * 1024u;
            const bool in = i < n;
            pv[k] = in ? prov[i] : make_uint4(0u, 0u, 0u, 0u);
            cn[k] = in ? counts[i] : 0u;
            sc[k] = in ? scores[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + (uint32_t)k * 1024u;
            if (i >= n) break;
            tm_best_record c;
            c.inlier_count = cn[k];
            c.score = sc[k];
            c.hyp_id = (pair_offset + pv[k].x * pair_stride) * limit + pv[k].w;
            if (record_better(c, b)) {
                b.inlier_count = c.inlier_count;
                b.score = c.score;
                b.hyp_id = c.hyp_id;
                bi = i;
            }
        }
    }
    /* wave reduction on (count, score, id, index), then the 16 wave winners in LDS */
    for (int off = 32; off > 0; off >>= 1) {
        tm_best_record c;
        c.inlier_count = __shfl_down(b.inlier_count, off);
        c.score = __shfl_down(b.score, off);
        c.hyp_id = __shfl_down(b.hyp_id, off);
        const uint32_t ci = __shfl_down(bi, off);
        if (record_better(c, b)) {
            b.inlier_count = c.inlier_count;
            b.score = c.score;
            b.hyp_id = c.hyp_id;
            bi = ci;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_rec[threadIdx.x >> 6].inlier_count = b.inlier_count;
        s_rec[threadIdx.x >> 6].score = b.score;
        s_rec[threadIdx.x >> 6].hyp_id = b.hyp_id;
        s_rec[threadIdx.x >> 6].pad[0] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        tm_best_record r = s_rec[0];
        for (int w = 1; w < 16; ++w)
            if (record_better(s_rec[w], r)) r = s_rec[w];
        const uint32_t i = r.pad[0];
        const bool have = r.hyp_id != 0xFFFFFFFFu;
        if (!have) {
            r.inlier_count = 0;
            r.score = 0.0;
        }
        for (int k = 0; k < 16; ++k) r.pose[k] = have ? poses[(size_t)i * 16 + k] : 0.f;
        for (int k = 0; k < 4; ++k) r.pad[k] = 0;
        *out = r;
    }
}

}  // namespace tmx

struct tm_comm {
    tm_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    tmx::rccl_comm_t comm = nullptr;
    tmx::dbuf<tm_best_record> d_local, d_all;
};

namespace tmx {
inline ctx_guard guard_of(const tm_comm* c) { return guard_of(c ? c->ctx : nullptr); }
}  // namespace tmx

using namespace tmx;

extern "C" {

int tm_comm_get_unique_id(unsigned char id[TM_COMM_ID_BYTES]) {
    TM_REQUIRE(id, TM_ERR_INVALID, "tm_comm_get_unique_id: null argument");
    TM_REQUIRE(rccl().ok, TM_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded");
    rccl_unique_id u;
    TM_RCCL_CHECK(rccl().GetUniqueId(&u));
    static_assert(sizeof(u) == TM_COMM_ID_BYTES, "unique id size");
    memcpy(id, &u, sizeof u);
    return TM_OK;
}

int tm_comm_create(tm_ctx* ctx, int rank, int world, const unsigned char id[TM_COMM_ID_BYTES], tm_comm** out) {
    TM_GUARD(ctx);
    TM_REQUIRE(ctx && out, TM_ERR_INVALID, "tm_comm_create: null argument");
    TM_REQUIRE(world >= 1 && rank >= 0 && rank < world, TM_ERR_INVALID, "tm_comm_create: bad rank / world");
    *out = nullptr;
    tm_comm* c = new tm_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    if (world > 1 || id) { /* a communicator of one rank is allowed (and exercises the same calls) */
        if (!id) {
            delete c;
            set_error("tm_comm_create: a unique id is needed for more than one rank");
            return TM_ERR_INVALID;
        }
        if (!rccl().ok) {
            delete c;
            set_error("RCCL (librccl.so.1) could not be loaded");
            return TM_ERR_UNSUPPORTED;
        }
        hipError_t e = hipSetDevice(ctx->device);
        rccl_unique_id u;
        memcpy(&u, id, sizeof u);
        int r = e == hipSuccess ? rccl().CommInitRank(&c->comm, world, u, rank) : 1;
        if (r != 0) {
            set_error("ncclCommInitRank failed: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
            delete c;
            return TM_ERR_HIP;
        }
    }
    *out = c;
    return TM_OK;
}

int tm_comm_destroy(tm_comm* c) {
    if (!c) return TM_OK;
    tmx::ctx_guard tm_guard_;
    if (ctx_alive(c->ctx)) {
        tm_guard_ = tmx::guard_of(c->ctx);
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        c->d_local.release();
        c->d_all.release();
        if (c->comm) (void)rccl().CommDestroy(c->comm);
    }
    delete c;
    return TM_OK;
}

int tm_comm_rank(const tm_comm* c) {
    TM_GUARD(c); return c ? c->rank : 0; }
int tm_comm_world(const tm_comm* c) {
    TM_GUARD(c); return c ? c->world : 1; }

int tm_best_of_batch_prov(tm_ctx* c, const uint32_t* d_counts, const double* d_scores, const float* d_poses16,
                          const uint32_t* d_provenance4, uint32_t n_hyp, uint32_t pair_offset, uint32_t pair_stride,
                          uint32_t query_limit, tm_best_record* d_best) {
    TM_GUARD(c);
    TM_REQUIRE(c && d_best, TM_ERR_INVALID, "tm_best_of_batch_prov: null argument");
    TM_REQUIRE((d_counts && d_scores && d_poses16 && d_provenance4) || n_hyp == 0, TM_ERR_INVALID,
               "tm_best_of_batch_prov: null device buffers");
    TM_REQUIRE(query_limit >= 1 && pair_stride >= 1, TM_ERR_INVALID, "tm_best_of_batch_prov: bad id parameters");
    TM_HIP_CHECK(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_best_prov, dim3(1), dim3(1024), 0, c->stream, d_counts, d_scores, d_poses16,
                       (const uint4*)d_provenance4, n_hyp, pair_offset, pair_stride, query_limit, d_best);
    TM_HIP_CHECK(hipGetLastError());
    return TM_OK;
}

/* RCCL usable in this process? (librccl.so.1 loaded and the entry points bound.) Ranks agree on the answer
 * BEFORE any of them enters the collective tm_comm_create, so that a rank without RCCL cannot leave the others
 * waiting inside ncclCommInitRank. */
int tm_comm_probe(void) {
    TM_REQUIRE(rccl().ok, TM_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded");
    return TM_OK;
}

/* device records in, host records out, in two halves: _begin queues the all-gather on the context's stream (no
 * host hop for the payload), the selection kernel and the copy of n_models x 96 bytes into pinned host memory,
 * records an event and returns; _end waits for that event only. Between the two the caller may queue further
 * work on the context (the next batch), which then overlaps the exchange. */
int tm_best_reduce_begin(tm_ctx* ctx, tm_comm* comm, const tm_best_record* d_local, uint32_t n_models, int* ticket) {
    TM_GUARD(ctx);
    TM_REQUIRE(ctx && d_local && ticket, TM_ERR_INVALID, "tm_best_reduce_begin: null argument");
    TM_REQUIRE(n_models >= 1, TM_ERR_INVALID, "tm_best_reduce_begin: no records");
    TM_REQUIRE(!comm || comm->ctx == ctx, TM_ERR_INVALID, "tm_best_reduce_begin: communicator of another context");
    *ticket = -1;
    TM_HIP_CHECK(hipSetDevice(ctx->device));
    /* any free slot, looked for from the one after the last hand-out (tickets ended out of order leave holes). The
     * capacity check comes before anything collective: ranks that make the same sequence of begin / end calls reach
     * it together, so none of them is left inside the all-gather. */
    int si = -1;
    for (int k = 0; k < kReduceSlots && si < 0; ++k)
        if (!ctx->red[(ctx->red_next + (uint32_t)k) % kReduceSlots].busy) si = (int)((ctx->red_next + (uint32_t)k) % kReduceSlots);
    TM_REQUIRE(si >= 0, TM_ERR_CAPACITY, "tm_best_reduce_begin: too many reductions in flight (tm_best_reduce_end first)");
    reduce_slot& slot = ctx->red[si];
    const size_t bytes = (size_t)n_models * sizeof(tm_best_record);
    if (slot.bytes < bytes) {
        if (slot.pinned) (void)hipHostFree(slot.pinned);
        slot.pinned = nullptr;
        slot.bytes = 0;
        TM_HIP_CHECK(hipHostMalloc(&slot.pinned, bytes, hipHostMallocDefault));
        slot.bytes = bytes;
    }
    if (!slot.ev) TM_HIP_CHECK(hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
    const uint32_t world = comm ? (uint32_t)comm->world : 1u;
    const tm_best_record* d_all = d_local;
    if (comm && comm->comm) {
        if (comm->d_all.n < (size_t)world * n_models) { /* growing frees the old buffer: let its readers finish */
            TM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            TM_FORWARD(comm->d_all.alloc((size_t)world * n_models));
        }
        TM_RCCL_CHECK(rccl().AllGather(d_local, comm->d_all.p, bytes, 1 /* ncclUint8 */, comm->comm, ctx->stream));
        d_all = comm->d_all.p;
    } else {
        TM_REQUIRE(world == 1, TM_ERR_INVALID, "tm_best_reduce: communicator without RCCL state");
    }
    if (ctx->red_out.n < n_models) {
        TM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        TM_FORWARD(ctx->red_out.alloc(n_models));
    }
    hipLaunchKernelGGL(k_best_select, dim3((n_models + 63) / 64), dim3(64), 0, ctx->stream, d_all, world, n_models,
                       ctx->red_out.p);
    TM_HIP_CHECK(hipGetLastError());
    TM_HIP_CHECK(hipMemcpyAsync(slot.pinned, ctx->red_out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TM_HIP_CHECK(hipEventRecord(slot.ev, ctx->stream));
    slot.n = n_models;
    slot.busy = true;
    slot.gen = (slot.gen + 1u) & 0x3FFFFFu; /* a ticket names the slot AND this use of it: good once */
    *ticket = si | (int)(slot.gen << 8);
    ctx->red_next = (uint32_t)si + 1u;
    return TM_OK;
}

int tm_best_reduce_end(tm_ctx* ctx, int ticket, tm_best_record* best) {
    TM_GUARD(ctx);
    TM_REQUIRE(ctx && best, TM_ERR_INVALID, "tm_best_reduce_end: null argument");
    const int si = ticket & 0xFF;
    TM_REQUIRE(ticket >= 0 && si < kReduceSlots && ctx->red[si].busy && ctx->red[si].gen == (uint32_t)(ticket >> 8),
               TM_ERR_INVALID, "tm_best_reduce_end: no reduction in flight under this ticket");
    reduce_slot& slot = ctx->red[si];
    slot.busy = false;
    TM_HIP_CHECK(hipEventSynchronize(slot.ev));
    memcpy(best, slot.pinned, (size_t)slot.n * sizeof(tm_best_record));
    return TM_OK;
}

/* both halves in one call */
int tm_best_reduce(tm_ctx* ctx, tm_comm* comm, const tm_best_record* d_local, uint32_t n_models,
                   tm_best_record* best) {
    TM_GUARD(ctx);
    TM_REQUIRE(best, TM_ERR_INVALID, "tm_best_reduce: null argument");
    int ticket = -1;
    TM_FORWARD(tm_best_reduce_begin(ctx, comm, d_local, n_models, &ticket));
    return tm_best_reduce_end(ctx, ticket, best);
}

/* host records in, host records out (the candidates of find_parallel's chunks): staged through the device
 * because RCCL moves device memory; all[r * n + k] = rank r's k-th record */
int tm_comm_allgather_records(tm_comm* comm, const tm_best_record* local, uint32_t n, tm_best_record* all) {
    TM_GUARD(comm);
    TM_REQUIRE(comm && local && all, TM_ERR_INVALID, "tm_comm_allgather_records: null argument");
    if (n == 0) return TM_OK;
    tm_ctx* ctx = comm->ctx;
    TM_HIP_CHECK(hipSetDevice(ctx->device));
    if (!comm->comm) {
        TM_REQUIRE(comm->world == 1, TM_ERR_INVALID, "tm_comm_allgather_records: communicator without RCCL state");
        memcpy(all, local, (size_t)n * sizeof(tm_best_record));
        return TM_OK;
    }
    TM_FORWARD(comm->d_local.ensure(n));
    TM_FORWARD(comm->d_all.ensure((size_t)comm->world * n));
    TM_HIP_CHECK(hipMemcpyAsync(comm->d_local.p, local, (size_t)n * sizeof(tm_best_record), hipMemcpyHostToDevice,
                                ctx->stream));
    TM_RCCL_CHECK(rccl().AllGather(comm->d_local.p, comm->d_all.p, (size_t)n * sizeof(tm_best_record), 1, comm->comm,
                                   ctx->stream));
    TM_HIP_CHECK(hipMemcpyAsync(all, comm->d_all.p, (size_t)comm->world * n * sizeof(tm_best_record),
                                hipMemcpyDeviceToHost, ctx->stream));
    TM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return TM_OK;
}

}  // extern "C"
