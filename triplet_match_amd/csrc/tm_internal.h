/* tm_internal.h — handles, error plumbing and device buffers shared by the .hip translation units. */
#ifndef TM_INTERNAL_H_
#define TM_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <new>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/tm_hip.h"
#include "tm_math.h"

struct tm_ctx;
namespace tmx {

void set_error(const char* fmt, ...);

#define TM_HIP_CHECK(expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            tmx::set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__,                  \
                          hipGetErrorString(_e));                                              \
            return TM_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)

#define TM_REQUIRE(cond, code, msg)  \
    do {                             \
        if (!(cond)) {               \
            tmx::set_error("%s", msg); \
            return code;             \
        }                            \
    } while (0)

#define TM_FORWARD(expr)          \
    do {                          \
        int _rc = (expr);         \
        if (_rc != TM_OK) return _rc; \
    } while (0)

/* Device memory of a context goes through its block cache (device_pool, tm_core.hip): a released block is kept and
 * handed to the next request of its size class instead of a hipFree (which waits for the whole device) and a hipMalloc
 * (~0.3 ms each; a new scene per frame made ~40 of them). Everything a context does runs on its one stream, so a block
 * that is reused is reused in stream order: work still queued on the old contents finishes first. The cache belongs
 * to the context a TM_GUARD names (thread-local, tmx::current_ctx()); outside any guard, or with the option
 * `pool_mib` = 0, the calls below are plain hipMalloc / hipFree. */
void* device_take(size_t bytes, size_t* capacity, tm_ctx** owner);
void device_give(void* p, size_t capacity, tm_ctx* owner);

/* RAII-less device buffer (freed explicitly by the owner's destroy) */
template <typename T>
struct dbuf {
    T* p = nullptr;
    size_t n = 0;
    size_t cap_bytes = 0;    /* what the block really holds (size class of the cache) */
    tm_ctx* owner = nullptr; /* whose cache it goes back to */
    int alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        p = (T*)device_take(count * sizeof(T), &cap_bytes, &owner);
        if (!p) return TM_ERR_HIP; /* error text set by device_take */
        n = count;
        return TM_OK;
    }
    int ensure(size_t count) { return (count <= n && p) ? TM_OK : alloc(count); }
    void release() {
        if (p) device_give(p, cap_bytes, owner);
        p = nullptr;
        n = 0;
        cap_bytes = 0;
        owner = nullptr;
    }
};

struct score_timing {
    float ms = 0.f;
    uint64_t evaluations = 0, passing = 0, hypotheses = 0, in_grid = 0, near = 0, items = 0;
};

/* Timed scoring launches are recorded without touching the host: an event pair and one row of device
 * counters per launch (ring of kTimedRing); tm_ctx_last_score_timing / tm_ctx_score_timing_totals
 * synchronise and read them when asked. */
constexpr int kTimedRing = 256;
constexpr int kTimedWords = 8; /* evaluations, passing, in-grid, near, staged items, 3 spare */

/* Switches of a context: cross-checks (a second kernel for the same result) and timing comparisons. Read ONCE
 * from the environment when the context is created (TM_<NAME> in upper case) and changed afterwards only through
 * tm_ctx_set_option; no call of the path reads the environment. None of them changes a result, except `ablate`,
 * which exists only in a library built with -DTM_TIMING_EXPERIMENTS. */
struct options {
    int ablate = 0;           /* TM_ABLATE (timing-experiments build only) */
    int debug_stats = 0;      /* per-launch stage counts on stderr */
    int debug_times = 0;      /* laps of tm_cloud_curvature */
    int debug_ed = 0;         /* pending counts of the early-drop prefix kernel */
    int score_grouped = 1;    /* 0: the per-hypothesis kernel (k_score_culled) for every launch */
    int ball_ungrouped = 0;   /* 1: k_score_culled for ball-restricted scoring */
    int score_item_wgs = 0;   /* persistent grid of the item kernel; 0 = what is resident */
    int item_sort = -1;       /* the items of the grouped kernel binned by the model region they land in: cells of
                                 2^item_sort voxels (2..6); -1 = 16-voxel cells for batches of 8,192 hypotheses and
                                 more against a voxel table of 256 MB and more (see score_device), 0 = never */
    int score_slot_mib = 256; /* byte budget of the grouped kernel's (group, chunk) slots per launch */
    int score_off32 = 1;      /* 0: 64-bit addresses for the grouped kernel's gathers also when every table is below 4 GiB */
    int near_metric = 1;      /* brick filters under a plane2 / cylinder metric: 0 never, 1 when at least a tenth of
                                 the grid's bricks lies beyond the threshold (auto), 2 always */
    int voxel_bruteforce = 0; /* exhaustive voxel NN kernel */
    int voxel_packet = 1;     /* 0: kd descent per voxel on per-thread stacks instead of one packet per brick */
    int voxel_seed = 1;       /* 0: the packet walk of the voxel grid starts without a candidate */
    int curv_groups = 1;      /* 0: the curvature pass's neighbour lists one thread per query (lists in LDS) instead of eight lanes per query */
    int model_host_build = 0; /* 1: tm_model_create sorts the cloud and builds a median kd-tree on the host (cross-check) */
    int ed_one_kernel = 0, ed_narrow = 0, ed_bruteforce = 0, ed_unfused = 0;
    int search_segment = -1;  /* hypotheses per segment of a search batch; -1 = 2048, 0 = no segments */
    int ed_bound_first = 1;   /* search steps: drop on the near-point bound before the exact first checkpoint (ed_args::bound_first) */
    int ed_prefix_list = 1;   /* 0: the first step of the early-drop rule walks the neighbourhood list (scan order)
                                 instead of the same positions in the scene's curve order */
    int ed_wave_rows = 8;     /* pending hypotheses per batch the continuation of a search step takes first (in visit
                                 order); the rest follows only for batches still without a sufficient one. 0 = all at once */
    int ed_prefix_group = 1;  /* 0: the first step of the rule one workgroup per hypothesis (k_ed_fused<true>) instead of
                                 up to 16 hypotheses of one neighbourhood per workgroup on interval reductions (1 = as many as
                                 keep the launch at ~4096 workgroups, 2..16 = that many) */
    int ed_ball_list = 1;     /* 0: the interval reductions walk the scene's tiles (k_ed_hist_wide) instead of the
                                 neighbourhoods' own lists (k_ed_hist_list) */
    int ed_hist = 1;          /* 0: the wide continuation keeps bit rows in global memory (k_ed_bits_wide / k_ed_walk /
                                 k_ed_score_rows) instead of the interval reductions of k_ed_hist_wide */
    int search_per_range = 0; /* 1: the batches of tm_search_batches run their segments one batch after the other */
    int search_step_check = 1; /* 0: every segment step of a merged round is queued; 1: the batches' stop words are read
                                  back after the first step and every third one, and a settled round queues no more */
    int pool_mib = 4096;      /* released device blocks the context keeps for reuse, in MiB (0: hipFree at once); 288 GB per
                                 device: 1 GiB was reached by 25 bench steps, and every release past it cost a hipFree */
};

/* released device blocks of one context by capacity (see device_take) */
struct device_pool {
    std::mutex mu;
    std::multimap<size_t, void*> free_blocks;
    size_t cached_bytes = 0;
    uint64_t hits = 0, misses = 0;
};

/* One tm_best_reduce in flight (tm_best_reduce_begin .. _end): the winners land in pinned host memory behind an
 * event, so the caller can queue the next batch before it waits. */
constexpr int kReduceSlots = 4;
struct reduce_slot {
    void* pinned = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
    uint32_t n = 0;
    uint32_t gen = 0; /* incremented per hand-out; part of the ticket, so a stale ticket is refused after reuse */
    bool busy = false;
};

}  // namespace tmx

struct tm_ctx {
    /* Every entry point that takes a context or a handle created from it holds this for the duration of the call
     * (TM_GUARD): calls from several threads on one context serialise instead of racing on its stream and work
     * buffers — the reference's workers call model::query / voxel_query / radius_search concurrently
     * (include/impl/scene.hpp:146-166,304). Recursive: entry points call each other. */
    mutable std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    bool timing = false;
    int timing_mode = 0;     /* tm_ctx_enable_timing: 0 off, 1 events + stage counters, 2 events only */
    tmx::options opt;        /* switches, see tmx::options */
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_copy = nullptr; /* behind an upload from a pinned staging buffer whose kernels the call does not wait for */
    /* one pinned staging buffer kept for the next scene (scene_pinned_ensure / scene_pinned_give): a scene per frame of a
     * stream otherwise pins and unpins a few megabytes two or three times per frame, a millisecond each */
    void* pinned_spare = nullptr;
    size_t pinned_spare_bytes = 0;
    std::vector<hipEvent_t> ev_ring;      /* 2 * kTimedRing, created on first use */
    uint64_t n_timed = 0;                 /* timed launches since tm_ctx_enable_timing(ctx, 1) */
    uint64_t timed_hyp[tmx::kTimedRing];  /* per ring row: hypotheses of the launch */
    uint64_t timed_evals_host[tmx::kTimedRing]; /* per ring row: evaluations known on the host (exhaustive kernel), else ~0 */
    tmx::dbuf<unsigned char> scratch;   /* rocprim temp storage */
    tmx::dbuf<unsigned long long> stats; /* device counters */
    tmx::reduce_slot red[tmx::kReduceSlots]; /* tm_best_reduce_begin / _end */
    tmx::dbuf<tm_best_record> red_out;        /* the selection kernel's output (stream-ordered reuse) */
    uint32_t red_next = 0;
    mutable tmx::device_pool pool;            /* released device blocks kept for reuse (tmx::device_take) */
};

struct tm_model {
    tm_ctx* ctx = nullptr;
    uint32_t n = 0;
    float resolution = 0.f;
    float distance_step_count = 0.f, angle_step = 0.f;
    int metric_kind = 0;       /* TM_TRAITS_*: which intrinsic_distance the scoring kernels use */
    float metric_radius = 0.f;
    std::vector<float> h_cloud; /* n*12 host copy (host-side query helpers) */
    /* geometry */
    bool grid_ready = false, voxels_ready = false, pairs_ready = false;
    float diameter = 0.f;
    float bbmin[3], bbmax[3];
    int extents[3] = {0, 0, 0};
    float to_voxel[16];
    tmx::grid_t grid;
    /* device SoA, stored in MORTON order of the positions ("new" indices): neighbouring scene points
     * gather neighbouring model records. The ABI speaks original indices; new2orig / orig2new map. */
    std::vector<uint32_t> h_new2orig, h_orig2new;
    tmx::dbuf<uint32_t> new2orig;
    tmx::dbuf<float4> vpos; /* xyz (NaN if non-finite), w = bits of the ORIGINAL index (voxel build tie-break) */
    /* kd-tree over the finite points for the voxel NN build. Node i = bounding box + links:
     * kd_lo[i] = (lo.xyz, bits a), kd_hi[i] = (hi.xyz, bits b); internal: a = left, b = right;
     * leaf: a = begin | 0x80000000, b = end. kd_pts = leaf-ordered (xyz, NEW index bits) */
    tmx::dbuf<float4> kd_lo, kd_hi;
    tmx::dbuf<float4> kd_pts;
    uint32_t kd_n_nodes = 0, kd_n_pts = 0;
    bool deep_tree = false; /* the tree is the device-built linear BVH (depth up to kLbvhMaxDepth): only walks with a
                               kLbvhStack-entry stack may descend it (the packet walk; never k_voxel_nn_kd's 24 entries) */
    tmx::dbuf<float4> pos;  /* xyz, w = (|tangent| > 0.7) ? 1 : 0 ; non-finite points keep NaN */
    tmx::dbuf<float4> ref;  /* tangent if is_tangent else normal */
    tmx::dbuf<float4> tan4; /* raw tangent xyz (pair build / hypothesis generation) */
    tmx::dbuf<float> pos3;  /* packed xyz, 12 B per point, for the culled scoring kernel (ref.w carries the flag) */
    tmx::dbuf<uint32_t> voxels; /* 4x4x4-bricked: [brick][z&3][y&3][x&3], entries are NEW indices */
    uint64_t voxel_count = 0;   /* dense count ex*ey*ez (ABI view) */
    uint64_t voxel_slots = 0;   /* bricks * 64 */
    /* 4x4x4-voxel bricks: quantised lower bound of the NN distance over the brick's voxel centres.
     * 1 byte per brick (L2-resident); lets the scoring kernels reject far-from-surface hits without
     * touching the big voxel table. */
    tmx::dbuf<uint32_t> brick_min_bits; /* f32 bits of min d2 per brick */
    tmx::dbuf<unsigned char> brick_code; /* floor(sqrt(min d2) / brick_q), clamped to 255 */
    /* one bit per super-brick: does any of its 64 bricks pass `code <= super_near_code` (the early-drop kernels' first,
     * LDS-resident filter); rebuilt when the threshold changes */
    tmx::dbuf<uint32_t> super_near;
    uint32_t super_near_code = 0xFFFFFFFFu;
    const unsigned char* super_near_src = nullptr;
    /* the same two filters under a projector's intrinsic_distance (plane2: 2-D, cylinder: wrap-around in u), built
     * by tm_model_set_metric: per brick the quantised minimum over its voxels of (a) the intrinsic distance from the
     * voxel centre to the voxel's STORED nearest point (what a query in that voxel is tested against) and (b) the
     * intrinsic distance from the centre to the nearest of ALL model points (a 1-Lipschitz field: tile tests) */
    tmx::dbuf<unsigned char> brick_code_metric_pt, brick_code_metric_tile;
    float metric_coord_max = 0.f; /* max |coordinate| of the model's points: rounding slack of the metric filters */
    uint64_t metric_code_hist[256] = {0}; /* bricks inside the grid per point-level code: how much the filter can remove */
    int bdim[3] = {0, 0, 0}; /* super-bricks (16^3 voxels = 64 bricks) per axis, see brick_of() */
    float brick_q = 0.f;
    std::vector<uint32_t> grid_subset; /* points the bounding box / diameter / grid are taken over; empty = all */
    /* pair table */
    std::vector<uint32_t> subset;
    float bmin[4], bmax[4];
    uint64_t n_entries = 0, n_valid = 0;
    uint32_t n_keys = 0;
    uint32_t r1 = 0, r2 = 0;            /* packing radices of k1, k2 */
    tmx::dbuf<uint32_t> seg_key;          /* packed key per distinct key (ascending) */
    tmx::dbuf<uint32_t> seg_begin;        /* n_keys + 1 */
    tmx::dbuf<uint2> entries;             /* (i,j) model indices, canonical order inside a segment */
    tmx::dbuf<uint32_t> ht_slot;          /* open addressing: segment index + 1, 0 = empty */
    tmx::dbuf<uint4> ht_key;              /* key stored in the slot */
    uint32_t ht_cap = 0;                 /* power of two */
    /* host mirrors for tm_model_query */
    std::vector<uint32_t> h_seg_key, h_seg_begin;
    std::vector<uint32_t> h_voxels_cache; /* lazily downloaded for tm_model_voxel_query */
};

namespace tmx {
/* Host mirror of a per-point int32 mask. An all-zero mask (every new scene starts with two) comes from calloc — fresh zero
 * pages that nothing touches until a helper reads them — instead of a 4 MB memset per million points, and says so
 * (all_zero), so that its device copy is a memset as well instead of an upload. */
class mask_mirror {
public:
    mask_mirror() = default;
    ~mask_mirror() { free(p_); }
    mask_mirror(const mask_mirror&) = delete;
    mask_mirror& operator=(const mask_mirror& o) {
        if (this != &o) assign(o.p_, o.p_ + o.n_);
        return *this;
    }
    void assign(size_t n, int32_t v) {
        free(p_);
        p_ = static_cast<int32_t*>(v == 0 ? calloc(n ? n : 1, 4) : malloc((n ? n : 1) * 4));
        if (!p_) throw std::bad_alloc();
        n_ = n;
        zero_ = v == 0;
        if (v != 0) std::fill(p_, p_ + n, v);
    }
    void assign(const int32_t* b, const int32_t* e) {
        const size_t n = (size_t)(e - b);
        int32_t* q = static_cast<int32_t*>(malloc((n ? n : 1) * 4));
        if (!q) throw std::bad_alloc();
        if (n) memcpy(q, b, n * 4);
        free(p_);
        p_ = q;
        n_ = n;
        zero_ = false;
    }
    void resize(size_t n) { /* content unspecified beyond the old size (callers overwrite it) */
        if (n == n_) return;
        int32_t* q = static_cast<int32_t*>(realloc(p_, (n ? n : 1) * 4));
        if (!q) throw std::bad_alloc();
        p_ = q;
        n_ = n;
        zero_ = false;
    }
    int32_t* data() { /* a writer may follow */
        zero_ = false;
        return p_;
    }
    const int32_t* data() const { return p_; }
    int32_t& operator[](size_t i) {
        zero_ = false;
        return p_[i];
    }
    const int32_t& operator[](size_t i) const { return p_[i]; }
    size_t size() const { return n_; }
    bool all_zero() const { return zero_; } /* known to be all zero (false = unknown) */

private:
    int32_t* p_ = nullptr;
    size_t n_ = 0;
    bool zero_ = false;
};
}  // namespace tmx

struct tm_scene {
    tm_ctx* ctx = nullptr;
    uint32_t n = 0;
    /* host mirror of the cloud (12 floats per point), filled from the device copy the first time a host-side helper
     * reads a row (tmx::scene_mirror): copying it at creation cost 6.5 ms of a 9 ms tm_scene_create per million
     * points, and the drop-in search never reads it */
    mutable std::vector<float> h_cloud;
    tmx::mask_mirror h_tmask, h_mask;
    tmx::dbuf<uint32_t> w_mask_idx; /* tm_scene_mask_add: the listed indices on the device */
    bool h_mask_stale = false; /* tm_scene_mask_add changed d_mask only: h_mask is refreshed on its next read (scene_host_mask) */
    /* A projected scene (tm_scene_create_projected) keeps its host mirror SPARSE: the host side of the search only
     * ever reads the surfels of tangent points (sample filters), so only those come back from the device — 1.7 MB
     * instead of 48 MB per projector at 1M points. h_tan_idx ascending, h_tan_rows 12 floats each; h_cloud stays
     * empty. tm_scene_download fetches the full cloud from the device on demand. */
    bool h_sparse = false;
    std::vector<uint32_t> h_tan_idx;
    std::vector<float> h_tan_rows;
    /* device copies of the inputs (original order) and the Morton order derived from them */
    tmx::dbuf<float> d_cloud;         /* n x 12 floats: the surfels as handed over */
    tmx::dbuf<int32_t> d_tmask, d_mask;
    tmx::dbuf<uint32_t> d_order;      /* sorted slot -> original index (independent of the mask) */
    tmx::dbuf<unsigned long long> d_keys; /* the sorted slots' curve keys (63 bits, an octree cell per 3-bit prefix step;
                                             non-finite points ~0): what tm_scene_curvature_subset builds its tree from */
    tmx::dbuf<unsigned char> d_bounds; /* tmx::scene_bounds (tm_scene_build.hip) */
    uint64_t n_live = 0;              /* finite and unmasked points (scene_live_count: read back on demand while live_pending) */
    bool live_pending = false;        /* a mask change recounted it on the device; nobody has asked since */
    /* original order (hypothesis generation addresses points by original index) */
    tmx::dbuf<float4> opos;  /* xyz, w = tangent_mask ? 1 : 0 */
    tmx::dbuf<float4> otan;  /* raw tangent xyz, w = masked ? 1 : 0 */
    tmx::dbuf<float4> oref;  /* tangent if tangent_mask else normal, w = tangent_mask */
    /* Morton-sorted order (scoring): 64 consecutive points form a tile with a bounding sphere */
    tmx::dbuf<float4> spos;       /* sorted xyz (masked / padding / non-finite -> NaN in x), w = bits of original index */
    tmx::dbuf<float> spos3;       /* sorted xyz packed 12 B per point (same NaN convention) */
    tmx::dbuf<float4> sref;       /* sorted ref (tangent or normal), w = tangent_mask ? 1 : 0 */
    tmx::dbuf<float4> tiles;      /* per tile: centre xyz, radius (negative = empty) */
    tmx::dbuf<float4> supers;     /* per super-tile (64 tiles = 4096 points): bounding sphere */
    tmx::dbuf<float4> subs;       /* per quarter tile (16 points): bounding sphere */
    tmx::dbuf<float4> chunks;     /* per chunk (512 points = 8 tiles): bounding sphere (grouped scoring) */
    uint32_t n_sorted = 0;        /* padded point count (multiple of the brute-force tile) */
    uint32_t n_tiles = 0;         /* n_sorted / 64 */
    float coord_max = 0.f;        /* max |coordinate| over the finite points */
    /* work buffers reused across calls */
    tmx::dbuf<float> w_poses;
    tmx::dbuf<uint32_t> w_counts;
    tmx::dbuf<unsigned long long> w_scores_fx;
    tmx::dbuf<double> w_scores;
    tmx::dbuf<uint32_t> w_u32a, w_u32b, w_u32c;
    tmx::dbuf<uint4> w_stats;
    tmx::dbuf<uint2> w_pairs;
    tmx::dbuf<uint32_t> w_masks;       /* grouped scoring: hypothesis mask per (group, chunk) */
    tmx::dbuf<uint32_t> w_items;       /* grouped scoring: slots with a non-zero mask, + the count */
    tmx::dbuf<float2> w_aux;           /* grouped scoring: per hypothesis (rnorm, abs_eps) */
    tmx::dbuf<uint32_t> w_item_bin, w_items_sorted, w_item_hist; /* grouped scoring: items binned by model region */
    tmx::dbuf<uint32_t> w_heads;       /* grouped scoring: per-XCD head words of the item hand-out */
    /* early-drop path (tm_earlydrop.hip) */
    tmx::dbuf<int32_t> ed_subset, ed_dropped;
    tmx::dbuf<uint32_t> ed_tests, ed_saved, ed_limit, ed_pos;
    tmx::dbuf<unsigned long long> ed_tail, ed_corr, ed_tilebits, ed_member;
    tmx::dbuf<uint32_t> ed_slot, ed_centres, ed_wprefix, ed_S;
    tmx::dbuf<uint32_t> d_tangent_idx; /* ascending indices of the points with tangent_mask set, + their number */
    uint32_t n_tangent = 0;
    bool tangent_idx_ready = false;
    tmx::dbuf<uint32_t> ed_tlist;      /* staging + CSR (rank, index) pairs of tm_scene_radius_tangent_lists */
    tmx::dbuf<uint2> ed_plist;         /* first-checkpoint positions of every neighbourhood in curve order */
    tmx::dbuf<float4> ed_plist_p;      /* the same entries' (x, y, z, sorted slot): what k_ed_prefix_group streams */
    tmx::dbuf<uint2> ed_blist;         /* every member of every neighbourhood in curve order (ed_ball::blist) */
    tmx::dbuf<uint32_t> ed_blist_n;
    tmx::dbuf<uint32_t> ed_plist_n;    /* their counts, then the cut-off B per slot */
    tmx::dbuf<uint32_t> ed_sort_scratch; /* k_pending_sort beyond its LDS rank sort: one bit per hypothesis + prefix counts */
    tmx::dbuf<uint32_t> ed_row_info;   /* per pending row: position among its batch's pending rows | batch << 20 */
    tmx::dbuf<unsigned long long> ed_hist; /* k_ed_hist_wide: interval counts / sums / minima per pending row */
    tmx::dbuf<uint32_t> ed_fb;      /* rows k_ed_hist_walk hands to the bit-array kernel, + one count per wave */
    tmx::dbuf<uint32_t> ed_pending; /* hypotheses the prefix kernel of the early-drop rule hands to the full kernel, + count */
    tmx::dbuf<float4> ed_cpos;
    /* tm_search_batch: pairs, provenance, slot of a pair's first point, centre per hypothesis, result record */
    tmx::dbuf<uint32_t> sb_ij, sb_pair_slot, sb_centres;
    tmx::dbuf<uint4> sb_prov;
    tmx::dbuf<tm_batch_result> sb_result;
    tmx::dbuf<unsigned char> w_active_sorted; /* subset flags in sorted order */
    tmx::dbuf<unsigned char> w_active;        /* subset flags in original order (tm_score_hypotheses) */
    std::vector<unsigned char> h_act;         /* host copy of the same, all zero between calls */
    tmx::dbuf<uint32_t> w_corr;            /* compacted (scene, model) correspondences, 2 x n */
    std::vector<uint32_t> extract_counts;  /* entries per row of w_corr after the last tm_extract_*_batch (tm_scene_mask_add_extracted) */
    tmx::dbuf<unsigned char> w_owner;      /* k_arb_claim: per scene point the number of the last candidate that claimed it */
    tmx::dbuf<uint32_t> w_kept_tan;        /* ... and per candidate (kept, tangent points among them) */
    tmx::dbuf<unsigned long long> w_words; /* membership bits of tm_scene_radius_subset */
    std::vector<unsigned long long> h_words;
    void* h_pinned = nullptr; /* pinned staging memory of tm_scene_radius_lists */
    size_t h_pinned_bytes = 0;
};

namespace tmx {
/* the context the calling thread is inside (innermost TM_GUARD), or null */
tm_ctx*& current_ctx_slot();
inline tm_ctx* current_ctx() { return current_ctx_slot(); }
/* holds the context's mutex and names it as the thread's current context for the scope */
class ctx_guard {
public:
    ctx_guard() = default;
    explicit ctx_guard(const tm_ctx* c) : lock_(c->mu), ctx_(const_cast<tm_ctx*>(c)), active_(true) {
        prev_ = current_ctx_slot();
        current_ctx_slot() = ctx_;
    }
    ctx_guard(ctx_guard&& o) noexcept : lock_(std::move(o.lock_)), ctx_(o.ctx_), prev_(o.prev_), active_(o.active_) {
        o.active_ = false;
    }
    ctx_guard& operator=(ctx_guard&& o) noexcept {
        if (this != &o) {
            leave();
            lock_ = std::move(o.lock_);
            ctx_ = o.ctx_;
            prev_ = o.prev_;
            active_ = o.active_;
            o.active_ = false;
        }
        return *this;
    }
    ctx_guard(const ctx_guard&) = delete;
    ctx_guard& operator=(const ctx_guard&) = delete;
    ~ctx_guard() { leave(); }

private:
    void leave() {
        if (active_) current_ctx_slot() = prev_;
        active_ = false;
        if (lock_.owns_lock()) lock_.unlock();
    }
    std::unique_lock<std::recursive_mutex> lock_;
    tm_ctx* ctx_ = nullptr;
    tm_ctx* prev_ = nullptr;
    bool active_ = false;
};
inline ctx_guard guard_of(const tm_ctx* c) { return c ? ctx_guard(c) : ctx_guard(); }
inline ctx_guard guard_of(const tm_model* m) { return guard_of(m ? m->ctx : nullptr); }
inline ctx_guard guard_of(const tm_scene* s) { return guard_of(s ? s->ctx : nullptr); }
}  // namespace tmx
/* first statement of an entry point: the handle's context stays locked until the call returns */
#define TM_GUARD(h) tmx::ctx_guard tm_guard_ = tmx::guard_of(h)

namespace tmx {
int scratch_ensure(tm_ctx* ctx, size_t bytes);
/* timed scoring launches [first, n_timed): synchronises the stream, sums events and device counters */
int read_timed(tm_ctx* ctx, uint64_t first, score_timing* sum, uint32_t* launches);
/* false once the context was destroyed or the process is exiting: destroy calls then free host memory only */
bool ctx_alive(tm_ctx* ctx);
/* tm_scene_build.hip: upload h_cloud / h_tmask / h_mask; derive every device array of the scene */
int scene_upload(tm_scene* s, const float* surfels = nullptr);
int scene_build(tm_scene* s, bool resort);
/* tm_model.hip: the linear BVH over points sorted by their curve keys (node format of the kd-tree kernels: lo.w / hi.w =
 * children, or leaf range begin | 0x80000000 / end). Deeper than a median kd-tree: a consumer needs a stack of
 * kLbvhStack entries. lo / hi: 2 * n_fin nodes, q0 / q1: n_fin work items each, cnt: 3 + kLbvhMaxDepth + 3 words
 * (cnt[0] = node count afterwards). Queued on the context's stream (with a few 4-byte read-backs in between). */
constexpr int kLbvhStack = 48, kLbvhMaxDepth = kLbvhStack - 2;
static_assert(kLbvhStack >= kLbvhMaxDepth + 2, "walks with kLbvhStack entries never overflow on a device-built tree");
void lbvh_build(tm_ctx* c, const unsigned long long* d_keys, const float4* d_pts, uint32_t n_fin, float4* d_lo,
                float4* d_hi, uint4* d_q0, uint4* d_q1, uint32_t* d_cnt);
void curve_points(tm_ctx* c, const float* d_cloud, const uint32_t* d_order, uint32_t n, float4* d_pts, uint32_t* d_n_fin);
/* tm_scene_build.hip: sorted curve keys + order of a raw device cloud (d_bounds: room for a tmx::scene_bounds, 64 bytes) */
int curve_order(tm_ctx* c, const float* d_cloud, uint32_t n, bool hilbert, void* d_bounds,
                dbuf<unsigned long long>& keys_out, dbuf<uint32_t>& order_out);
/* ascending indices of the points with tangent_mask set (s->d_tangent_idx, s->n_tangent), built once per scene */
int scene_tangent_index(tm_scene* s);
/* the scene's pinned staging buffer (s->h_pinned) with room for at least `need` bytes; grown with a quarter of head
 * room and never below 4 MiB, so the read-backs of a search do not reallocate it; taken from the context's spare when that fits */
/* a scene's staging buffer goes back to its context (the larger of the two stays, the other is unpinned); callers hold
 * the context's mutex (TM_GUARD) */
inline void scene_pinned_give(tm_ctx* c, void* p, size_t bytes) {
    if (!p) return;
    if (c && bytes > c->pinned_spare_bytes) {
        if (c->pinned_spare) (void)hipHostFree(c->pinned_spare);
        c->pinned_spare = p;
        c->pinned_spare_bytes = bytes;
        return;
    }
    (void)hipHostFree(p);
}
inline int scene_pinned_ensure(tm_scene* s, size_t need) {
    if (s->h_pinned_bytes >= need && s->h_pinned) return TM_OK;
    tm_ctx* c = s->ctx;
    scene_pinned_give(c, s->h_pinned, s->h_pinned_bytes);
    s->h_pinned = nullptr;
    s->h_pinned_bytes = 0;
    if (c->pinned_spare && c->pinned_spare_bytes >= need) { /* what the last scene of this context left */
        s->h_pinned = c->pinned_spare;
        s->h_pinned_bytes = c->pinned_spare_bytes;
        c->pinned_spare = nullptr;
        c->pinned_spare_bytes = 0;
        return TM_OK;
    }
    const size_t bytes = std::max(need + need / 4, (size_t)4 << 20);
    TM_HIP_CHECK(hipHostMalloc(&s->h_pinned, bytes, hipHostMallocDefault));
    s->h_pinned_bytes = bytes;
    return TM_OK;
}
/* fills s->h_cloud from the device copy if it is still empty (dense mirrors only); TM_OK or a HIP error */
int scene_mirror(const tm_scene* s);
/* host mirror row of point i (12 floats), or null for a point a sparse mirror does not hold */
const float* scene_row(const tm_scene* s, uint32_t i);
int scene_host_mask(const tm_scene* s);
/* n_live, fetched from the device if a mask change left it there (a search changes the mask twice and never asks) */
int scene_live_count(const tm_scene* s, uint64_t* live);
}

#endif /* TM_INTERNAL_H_ */
