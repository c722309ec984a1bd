/* tm_core.hip — context, error text, host-side scalar API and the batched descriptor kernel. */
#include "tm_internal.h"

#include <cctype>
#include <cstdlib>
#include <mutex>
#include <set>

namespace tmx {

static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
}

int scratch_ensure(tm_ctx* ctx, size_t bytes) { return ctx->scratch.ensure(bytes); }

/* one thread per (a,b) pair: descriptor, key and hash — the device twin of tm_feature /
 * tm_discretize_feature / tm_hash_key */
__global__ void k_features_batch(const float* __restrict__ surfels, const uint2* __restrict__ ab,
                                 uint32_t n_pairs, int have_bounds, float lo0, float hi0,
                                 uint32_t steps, float angle_step, float4* __restrict__ feats,
                                 uint4* __restrict__ keys, uint32_t* __restrict__ hashes) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pairs) return;
    uint2 p = ab[t];
    const float* s0 = surfels + (size_t)p.x * 12;
    const float* s1 = surfels + (size_t)p.y * 12;
    feat4 f = pair_feature(make3(s0[0], s0[1], s0[2]), make3(s0[9], s0[10], s0[11]),
                           make3(s1[0], s1[1], s1[2]), make3(s1[9], s1[10], s1[11]));
    feats[t] = make_float4(f.f0, f.f1, f.f2, f.f3);
    if (have_bounds) {
        key4 k = feature_key(f, lo0, hi0, steps, angle_step);
        keys[t] = make_uint4(k.k0, k.k1, k.k2, k.k3);
        hashes[t] = murmur_key4(k.k0, k.k1, k.k2, k.k3);
    }
}

}  // namespace tmx

using namespace tmx;

extern "C" {

const char* tm_last_error(void) { return g_error.c_str(); }
#ifdef TM_TIMING_EXPERIMENTS
const char* tm_version(void) { return "triplet_match_amd 0.4 (gfx950) +timing-experiments: NOT a product build"; }
#else
const char* tm_version(void) { return "triplet_match_amd 0.4 (gfx950)"; }
#endif

}  // extern "C"

namespace tmx {
/* live contexts: a model / scene destroyed after its context (garbage collection in arbitrary order at
 * interpreter exit) must not touch the freed context or the runtime */
static std::mutex g_ctx_mutex;
static std::set<tm_ctx*> g_ctx_live;
static bool g_exiting = false;
bool ctx_alive(tm_ctx* c) {
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    return !g_exiting && g_ctx_live.count(c) != 0;
}
static void mark_exit() {
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    g_exiting = true;
}
}  // namespace tmx

namespace tmx {
tm_ctx*& current_ctx_slot() {
    static thread_local tm_ctx* cur = nullptr;
    return cur;
}

/* size classes of the block cache: three significant bits (at most 12.5 % above the request), 4 KiB at least */
static size_t pool_class(size_t bytes) {
    if (bytes <= 4096) return 4096;
    int top = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (top - 3);
    return (bytes + step - 1) & ~(step - 1);
}

static void pool_trim_locked(device_pool& pool, size_t keep) {
    while (pool.cached_bytes > keep && !pool.free_blocks.empty()) {
        auto it = std::prev(pool.free_blocks.end()); /* largest first: the least likely to be asked for again */
        pool.cached_bytes -= it->first;
        (void)hipFree(it->second);
        pool.free_blocks.erase(it);
    }
}

/* An allocation failed: give back what EVERY live context of this process holds in its cache (contexts of other
 * threads may sit on gigabytes of released blocks while this one runs out), not only the caller's. */
static void pool_trim_all() {
    std::lock_guard<std::mutex> lk(g_ctx_mutex);
    if (g_exiting) return;
    for (tm_ctx* other : g_ctx_live) {
        std::lock_guard<std::mutex> pl(other->pool.mu);
        pool_trim_locked(other->pool, 0);
    }
}

void* device_take(size_t bytes, size_t* capacity, tm_ctx** owner) {
    tm_ctx* c = current_ctx();
    if (bytes == 0) bytes = 1;
    *owner = nullptr;
    *capacity = bytes;
    void* p = nullptr;
    if (c && c->opt.pool_mib > 0) {
        const size_t cap = pool_class(bytes);
        {
            std::lock_guard<std::mutex> lk(c->pool.mu);
            auto it = c->pool.free_blocks.find(cap);
            if (it != c->pool.free_blocks.end()) {
                p = it->second;
                c->pool.cached_bytes -= cap;
                c->pool.free_blocks.erase(it);
                ++c->pool.hits;
            } else {
                ++c->pool.misses;
            }
        }
        if (!p) {
            hipError_t e = hipMalloc(&p, cap);
            if (e != hipSuccess) { /* give the caches back to the device and try once more */
                (void)hipGetLastError();
                pool_trim_all();
                e = hipMalloc(&p, cap);
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                set_error("hipMalloc(%zu bytes) failed: %s", cap, hipGetErrorString(e));
                return nullptr;
            }
        }
        *capacity = cap;
        *owner = c;
        return p;
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pool_trim_all();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

void device_give(void* p, size_t capacity, tm_ctx* owner) {
    if (!p) return;
    /* back to its context's cache while that context lives and keeps one; a block larger than half the budget is
     * not worth holding on to (a voxel table). The liveness check and the insertion happen under g_ctx_mutex, which
     * tm_ctx_destroy takes before it unlists and deletes a context: a block released by another thread (a local dbuf,
     * garbage collection in arbitrary order) can no longer reach a context that is being torn down. */
    if (owner) {
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        if (!g_exiting && g_ctx_live.count(owner) != 0 && owner->opt.pool_mib > 0) {
            const size_t limit = (size_t)owner->opt.pool_mib << 20;
            if (capacity <= limit / 2) {
                std::lock_guard<std::mutex> pl(owner->pool.mu);
                owner->pool.free_blocks.emplace(capacity, p);
                owner->pool.cached_bytes += capacity;
                /* past the budget: down to three quarters of it in one go — trimmed to the brim, every further release
                 * evicted a block (a hipFree waits for the device) and the next request of that size missed: a voxel
                 * grid build took 30 ms instead of 9 once 25 bench steps had filled the cache */
                if (owner->pool.cached_bytes > limit) pool_trim_locked(owner->pool, limit - limit / 4);
                return;
            }
        }
    }
    (void)hipFree(p);
}

struct opt_entry {
    const char* name;
    int options::*field;
    int lo, hi;
};
static const opt_entry kOptions[] = {
    {"ablate", &options::ablate, 0, 255},
    {"debug_stats", &options::debug_stats, 0, 1},
    {"debug_times", &options::debug_times, 0, 1},
    {"debug_ed", &options::debug_ed, 0, 1},
    {"score_grouped", &options::score_grouped, 0, 1},
    {"ball_ungrouped", &options::ball_ungrouped, 0, 1},
    {"score_item_wgs", &options::score_item_wgs, 0, 1 << 20},
    {"item_sort", &options::item_sort, -1, 6},
    {"score_slot_mib", &options::score_slot_mib, 1, 1 << 16},
    {"near_metric", &options::near_metric, 0, 2},
    {"voxel_bruteforce", &options::voxel_bruteforce, 0, 1},
    {"voxel_packet", &options::voxel_packet, 0, 1},
    {"voxel_seed", &options::voxel_seed, 0, 1},
    {"curv_groups", &options::curv_groups, 0, 1},
    {"model_host_build", &options::model_host_build, 0, 1},
    {"ed_one_kernel", &options::ed_one_kernel, 0, 1},
    {"ed_narrow", &options::ed_narrow, 0, 1},
    {"ed_bruteforce", &options::ed_bruteforce, 0, 1},
    {"ed_unfused", &options::ed_unfused, 0, 1},
    {"search_segment", &options::search_segment, -1, 1 << 24},
    {"ed_bound_first", &options::ed_bound_first, 0, 1},
    {"search_per_range", &options::search_per_range, 0, 1},
    {"search_step_check", &options::search_step_check, 0, 1},
    {"ed_wave_rows", &options::ed_wave_rows, 0, 1 << 19},
    {"ed_prefix_list", &options::ed_prefix_list, 0, 1},
    {"score_off32", &options::score_off32, 0, 1},
    {"ed_hist", &options::ed_hist, 0, 1},
    {"ed_ball_list", &options::ed_ball_list, 0, 1},
    {"ed_prefix_group", &options::ed_prefix_group, 0, 16},
    {"pool_mib", &options::pool_mib, 0, 1 << 20},
};
}  // namespace tmx

extern "C" {

int tm_ctx_set_option(tm_ctx* c, const char* name, int value) {
    TM_GUARD(c);
    TM_REQUIRE(c && name, TM_ERR_INVALID, "tm_ctx_set_option: null argument");
    for (const opt_entry& o : kOptions) {
        if (strcmp(o.name, name) != 0) continue;
        if (value < o.lo || value > o.hi) {
            set_error("tm_ctx_set_option: %s = %d is out of range [%d, %d]", name, value, o.lo, o.hi);
            return TM_ERR_INVALID;
        }
#ifndef TM_TIMING_EXPERIMENTS
        if (o.field == &options::ablate && value != 0) {
            set_error("ablate: timing experiments return wrong results and are not compiled into this library "
                      "(build with -DTM_TIMING_EXPERIMENTS: python -m triplet_match_amd.build --timing-experiments)");
            return TM_ERR_UNSUPPORTED;
        }
#endif
        c->opt.*(o.field) = value;
        if (o.field == &options::pool_mib) { /* a smaller budget takes effect at once */
            std::lock_guard<std::mutex> lk(c->pool.mu);
            pool_trim_locked(c->pool, (size_t)value << 20);
        }
        return TM_OK;
    }
    set_error("tm_ctx_set_option: unknown option '%s'", name);
    return TM_ERR_INVALID;
}

int tm_ctx_get_option(const tm_ctx* c, const char* name, int* value) {
    TM_GUARD(c);
    TM_REQUIRE(c && name && value, TM_ERR_INVALID, "tm_ctx_get_option: null argument");
    for (const opt_entry& o : kOptions)
        if (strcmp(o.name, name) == 0) {
            *value = c->opt.*(o.field);
            return TM_OK;
        }
    set_error("tm_ctx_get_option: unknown option '%s'", name);
    return TM_ERR_INVALID;
}

int tm_ctx_create(int device_id, tm_ctx** out) {
    TM_REQUIRE(out, TM_ERR_INVALID, "tm_ctx_create: null output");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return TM_ERR_NO_DEVICE;
    }
    TM_REQUIRE(device_id >= 0 && device_id < count, TM_ERR_INVALID, "tm_ctx_create: bad device id");
    TM_HIP_CHECK(hipSetDevice(device_id));
    tm_ctx* c = new tm_ctx();
    c->device = device_id;
    /* the one place the environment is read — before anything is allocated, so a bad value leaves nothing behind */
    for (const opt_entry& o : kOptions) {
        std::string env = "TM_";
        for (const char* p = o.name; *p; ++p) env += (char)toupper((unsigned char)*p);
        if (const char* v = getenv(env.c_str())) {
            const int value = *v ? atoi(v) : 1;
#ifndef TM_TIMING_EXPERIMENTS
            if (o.field == &options::ablate) { /* a leftover benchmarking variable must not take the product down */
                if (value != 0)
                    fprintf(stderr, "libtm_hip: %s=%d ignored (stage ablation exists only in the timing-experiments build)\n",
                            env.c_str(), value);
                continue;
            }
#endif
            int rc = tm_ctx_set_option(c, o.name, value);
            if (rc != TM_OK) {
                delete c;
                return rc;
            }
        }
    }
    auto undo = [&](int rc) {
        if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        c->stats.release();
        delete c;
        return rc;
    };
    hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreate(&c->ev0);
    if (he == hipSuccess) he = hipEventCreate(&c->ev1);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming);
    if (he != hipSuccess) {
        set_error("tm_ctx_create: stream / event creation failed: %s", hipGetErrorString(he));
        return undo(TM_ERR_HIP);
    }
    if (int rc = c->stats.alloc((size_t)kTimedRing * kTimedWords)) return undo(rc);
    {
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        static bool registered = false;
        if (!registered) {
            atexit(mark_exit);
            registered = true;
        }
        g_ctx_live.insert(c);
    }
    *out = c;
    return TM_OK;
}

int tm_ctx_destroy(tm_ctx* c) {
    if (!c || !ctx_alive(c)) return TM_OK;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        g_ctx_live.erase(c);
    }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->scratch.release();
    c->stats.release();
    c->red_out.release();
    {
        std::lock_guard<std::mutex> lk(c->pool.mu);
        pool_trim_locked(c->pool, 0);
    }
    for (reduce_slot& r : c->red) {
        if (r.pinned) (void)hipHostFree(r.pinned);
        if (r.ev) (void)hipEventDestroy(r.ev);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->pinned_spare) (void)hipHostFree(c->pinned_spare);
    for (hipEvent_t e : c->ev_ring) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TM_OK;
}

int tm_ctx_pool_stats(const tm_ctx* c, uint64_t* cached_bytes, uint64_t* reused, uint64_t* allocated) {
    TM_GUARD(c);
    TM_REQUIRE(c, TM_ERR_INVALID, "null ctx");
    std::lock_guard<std::mutex> lk(c->pool.mu);
    if (cached_bytes) *cached_bytes = c->pool.cached_bytes;
    if (reused) *reused = c->pool.hits;
    if (allocated) *allocated = c->pool.misses;
    return TM_OK;
}

int tm_ctx_synchronize(tm_ctx* c) {
    TM_GUARD(c);
    TM_REQUIRE(c, TM_ERR_INVALID, "null ctx");
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return TM_OK;
}

void* tm_ctx_stream(tm_ctx* c) { return c ? (void*)c->stream : nullptr; }

int tm_device_alloc(tm_ctx* c, uint64_t bytes, void** d_ptr) {
    TM_GUARD(c);
    TM_REQUIRE(c && d_ptr, TM_ERR_INVALID, "tm_device_alloc: null argument");
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_HIP_CHECK(hipMalloc(d_ptr, bytes ? bytes : 1));
    return TM_OK;
}
int tm_device_free(tm_ctx* c, void* d_ptr) {
    TM_GUARD(c);
    TM_REQUIRE(c, TM_ERR_INVALID, "null ctx");
    if (d_ptr) TM_HIP_CHECK(hipFree(d_ptr));
    return TM_OK;
}
int tm_memcpy_h2d(tm_ctx* c, void* d_dst, const void* src, uint64_t bytes) {
    TM_GUARD(c);
    TM_REQUIRE(c && d_dst && src, TM_ERR_INVALID, "tm_memcpy_h2d: null argument");
    TM_HIP_CHECK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return TM_OK;
}
int tm_memcpy_d2h(tm_ctx* c, void* dst, const void* d_src, uint64_t bytes) {
    TM_GUARD(c);
    TM_REQUIRE(c && dst && d_src, TM_ERR_INVALID, "tm_memcpy_d2h: null argument");
    TM_HIP_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return TM_OK;
}

/* a stopwatch on the context's stream (HIP events): everything queued between the two calls */
int tm_ctx_time_begin(tm_ctx* c) {
    TM_GUARD(c);
    TM_REQUIRE(c, TM_ERR_INVALID, "null ctx");
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
    return TM_OK;
}
int tm_ctx_time_end(tm_ctx* c, float* ms) {
    TM_GUARD(c);
    TM_REQUIRE(c && ms, TM_ERR_INVALID, "tm_ctx_time_end: null argument");
    TM_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
    TM_HIP_CHECK(hipEventSynchronize(c->ev1));
    TM_HIP_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return TM_OK;
}

int tm_ctx_enable_timing(tm_ctx* c, int on) {
    TM_GUARD(c);
    TM_REQUIRE(c, TM_ERR_INVALID, "null ctx");
    if (on && c->ev_ring.empty()) {
        TM_HIP_CHECK(hipSetDevice(c->device));
        c->ev_ring.resize(2 * (size_t)kTimedRing, nullptr);
        for (hipEvent_t& e : c->ev_ring) TM_HIP_CHECK(hipEventCreate(&e));
    }
    if (on) c->n_timed = 0; /* a new measurement window */
    c->timing = on != 0;
    c->timing_mode = on == 2 ? 2 : (on ? 1 : 0);
    return TM_OK;
}

}  // extern "C"

/* reads the rows [first, n_timed) of the ring (at most the last kTimedRing launches) */
int tmx::read_timed(tm_ctx* c, uint64_t first, score_timing* sum, uint32_t* launches) {
    *sum = score_timing();
    *launches = 0;
    if (c->n_timed == 0) return TM_OK;
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> rows((size_t)kTimedRing * kTimedWords);
    TM_HIP_CHECK(hipMemcpy(rows.data(), c->stats.p, rows.size() * 8, hipMemcpyDeviceToHost));
    if (c->n_timed - first > (uint64_t)kTimedRing) first = c->n_timed - kTimedRing;
    for (uint64_t k = first; k < c->n_timed; ++k) {
        const uint32_t row = (uint32_t)(k % kTimedRing);
        float ms = 0.f;
        TM_HIP_CHECK(hipEventElapsedTime(&ms, c->ev_ring[2 * row], c->ev_ring[2 * row + 1]));
        const unsigned long long* r = rows.data() + (size_t)row * kTimedWords;
        sum->ms += ms;
        sum->evaluations += c->timed_evals_host[row] != ~0ull ? c->timed_evals_host[row] : r[0];
        sum->passing += r[1];
        sum->in_grid += r[2];
        sum->near += r[3];
        sum->items += r[4];
        sum->hypotheses += c->timed_hyp[row];
        ++*launches;
    }
    return TM_OK;
}

extern "C" {

int tm_ctx_last_score_timing(tm_ctx* c, float* ms, uint64_t* evaluations, uint64_t* passing,
                             uint64_t* hypotheses) {
    TM_GUARD(c);
    TM_REQUIRE(c, TM_ERR_INVALID, "null ctx");
    score_timing t;
    uint32_t n = 0;
    TM_FORWARD(read_timed(c, c->n_timed ? c->n_timed - 1 : 0, &t, &n));
    if (ms) *ms = t.ms;
    if (evaluations) *evaluations = t.evaluations;
    if (passing) *passing = t.passing;
    if (hypotheses) *hypotheses = t.hypotheses;
    return TM_OK;
}

int tm_ctx_score_timing_totals(tm_ctx* c, uint32_t* launches, float* total_ms, uint64_t totals6[6]) {
    TM_GUARD(c);
    TM_REQUIRE(c, TM_ERR_INVALID, "null ctx");
    score_timing t;
    uint32_t n = 0;
    TM_FORWARD(read_timed(c, 0, &t, &n));
    if (launches) *launches = n;
    if (total_ms) *total_ms = t.ms;
    if (totals6) {
        totals6[0] = t.evaluations;
        totals6[1] = t.passing;
        totals6[2] = t.in_grid;
        totals6[3] = t.near;
        totals6[4] = t.hypotheses;
        totals6[5] = t.items;
    }
    return TM_OK;
}

/* ---- host scalar API ---- */
uint32_t tm_murmur3_32(const uint32_t* words, int n_words, uint32_t seed) {
    return murmur_words(words, n_words, seed);
}
uint32_t tm_hash_key(const uint32_t* key, int dim) { return murmur_words(key, dim, kMurmurSeed); }
uint32_t tm_discretize_range(float v, float lo, float span, uint32_t steps) {
    return quantize_range(v, lo, span, steps);
}
uint32_t tm_discretize_step(float v, float step) { return quantize_step(v, step); }
float tm_atan2f(float y, float x) { return atan2_def(y, x); }

void tm_feature(const float* s0, const float* s1, float out[4]) {
    feat4 f = pair_feature(make3(s0[0], s0[1], s0[2]), make3(s0[9], s0[10], s0[11]),
                           make3(s1[0], s1[1], s1[2]), make3(s1[9], s1[10], s1[11]));
    out[0] = f.f0;
    out[1] = f.f1;
    out[2] = f.f2;
    out[3] = f.f3;
}
void tm_discretize_feature(const float f[4], const float bmin[4], const float bmax[4],
                           float distance_step_count, float angle_step, uint32_t key[4]) {
    feat4 ff = {f[0], f[1], f[2], f[3]};
    key4 k = feature_key(ff, bmin[0], bmax[0], (uint32_t)distance_step_count, angle_step);
    key[0] = k.k0;
    key[1] = k.k1;
    key[2] = k.k2;
    key[3] = k.k3;
}
int tm_valid(const float f[4], const float bmin[4], const float bmax[4]) {
    feat4 ff = {f[0], f[1], f[2], f[3]};
    return feature_valid(ff, bmin[0], bmax[0]) ? 1 : 0;
}
void tm_valid_bounds(const float bmin[4], const float bmax[4], float min_rel, float max_rel,
                     float out_min[4], float out_max[4]) {
    for (int d = 0; d < 4; ++d) {
        out_min[d] = bmin[d];
        out_max[d] = bmax[d];
    }
    float s0 = bmax[0] - bmin[0], s3 = bmax[3] - bmin[3];
    out_min[0] = bmin[0] + min_rel * s0;
    out_max[0] = bmin[0] + max_rel * s0;
    out_min[3] = bmin[3] + min_rel * s3;
    out_max[3] = bmin[3] + max_rel * s3;
}
void tm_base_transform(const float si[3], const float sj[3], const float st[3], const float ti[3],
                       const float tj[3], const float tt[3], float pose16[16]) {
    base_transform(make3(si[0], si[1], si[2]), make3(sj[0], sj[1], sj[2]), make3(st[0], st[1], st[2]),
                   make3(ti[0], ti[1], ti[2]), make3(tj[0], tj[1], tj[2]), make3(tt[0], tt[1], tt[2]),
                   pose16);
}
uint32_t tm_early_drop_upper(uint32_t tried, uint32_t subset_size, uint32_t corrs) {
    return early_drop_upper(tried, subset_size, corrs);
}

int tm_features_batch(tm_ctx* c, const float* surfels, uint32_t n_points, const uint32_t* ab_pairs,
                      uint32_t n_pairs, const float bmin[4], const float bmax[4],
                      float distance_step_count, float angle_step, float* features4,
                      uint32_t* keys4, uint32_t* hashes) {
    TM_GUARD(c);
    TM_REQUIRE(c && surfels && ab_pairs && features4, TM_ERR_INVALID, "tm_features_batch: null argument");
    if (n_pairs == 0) return TM_OK;
    for (uint32_t i = 0; i < 2 * n_pairs; ++i)
        TM_REQUIRE(ab_pairs[i] < n_points, TM_ERR_INVALID, "tm_features_batch: index out of range");
    int have_bounds = (bmin && bmax && keys4 && hashes) ? 1 : 0;
    TM_HIP_CHECK(hipSetDevice(c->device));
    dbuf<float> d_s;
    dbuf<uint2> d_ab;
    dbuf<float4> d_f;
    dbuf<uint4> d_k;
    dbuf<uint32_t> d_h;
    int rc = TM_OK;
    auto cleanup = [&]() {
        d_s.release();
        d_ab.release();
        d_f.release();
        d_k.release();
        d_h.release();
    };
    if ((rc = d_s.alloc((size_t)n_points * 12)) || (rc = d_ab.alloc(n_pairs)) ||
        (rc = d_f.alloc(n_pairs)) || (rc = d_k.alloc(n_pairs)) || (rc = d_h.alloc(n_pairs))) {
        cleanup();
        return rc;
    }
    hipError_t e;
    e = hipMemcpyAsync(d_s.p, surfels, (size_t)n_points * 48, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_ab.p, ab_pairs, (size_t)n_pairs * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_features_batch, dim3((n_pairs + 255) / 256), dim3(256), 0, c->stream,
                           d_s.p, d_ab.p, n_pairs, have_bounds, have_bounds ? bmin[0] : 0.f,
                           have_bounds ? bmax[0] : 0.f, (uint32_t)distance_step_count, angle_step,
                           d_f.p, d_k.p, d_h.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(features4, d_f.p, (size_t)n_pairs * 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && have_bounds) {
        e = hipMemcpyAsync(keys4, d_k.p, (size_t)n_pairs * 16, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(hashes, d_h.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) {
        set_error("tm_features_batch: %s", hipGetErrorString(e));
        return TM_ERR_HIP;
    }
    return TM_OK;
}

/* deterministic order: inlier count desc, score desc, hypothesis id asc */
int tm_best_select(const tm_best_record* records, uint32_t n, tm_best_record* best) {
    TM_REQUIRE(records && best, TM_ERR_INVALID, "tm_best_select: null argument");
    tm_best_record b;
    memset(&b, 0, sizeof(b));
    b.hyp_id = 0xFFFFFFFFu;
    bool have = false;
    for (uint32_t i = 0; i < n; ++i) {
        const tm_best_record& r = records[i];
        if (r.hyp_id == 0xFFFFFFFFu) continue;
        bool better = !have || r.inlier_count > b.inlier_count ||
                      (r.inlier_count == b.inlier_count &&
                       (r.score > b.score || (r.score == b.score && r.hyp_id < b.hyp_id)));
        if (better) {
            b = r;
            have = true;
        }
    }
    *best = b;
    return TM_OK;
}

}  // extern "C"
