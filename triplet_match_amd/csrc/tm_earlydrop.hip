/*
 * tm_earlydrop.hip — project_ WITH its early-drop rule (reference include/impl/scene.hpp:420-426,
 * 492-506), evaluated exactly for a batch of poses.
 *
 * The rule is sequential in the subset's scan order: checkpoint k is tested at the first unmasked,
 * in-grid point whose 1-based position is >= tests[k] (points that `continue` at :434 / :458-460
 * bypass the test), with the correspondence count up to and including that point. Three passes:
 *   1. k_ed_bits_culled  one workgroup per hypothesis walks the Morton-sorted scene, skips every tile whose
 *                  sphere cannot reach the voxel box, and sets the in-grid bit and the correspondence
 *                  bit of each remaining point at its position in the scan order (64-bit atomicOr;
 *                  pos_of_orig maps a scene point to its position). Subsets that name a point twice
 *                  take k_ed_bits instead: one thread per (position, hypothesis), wave ballots;
 *   2. k_ed_walk   one wave per hypothesis walks the words, finds each checkpoint's test position
 *                  and applies the hypergeometric bound (SURVEY Q8 positive form);
 *   3. k_ed_score  score terms of the correspondences up to the stop position.
 * This reproduces the reference's outcome (count, dropped checkpoint, saved, score); it is the parity
 * path for "early-drop as shipped", not a work-saving prefilter.
 */
#include <cstring>

#include "tm_score.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>
#include <unordered_map>
#include <vector>

namespace tmx {

/* "ball mode": every hypothesis scans its own neighbourhood nn(centre) = radius_search_inclusive(radius,
 * centre) in ascending index order (what find_in_subset hands to project_, include/impl/scene.hpp:273,326).
 * Hypotheses name their centre by slot; per slot: membership bits by original index, the number of members
 * before each 64-bit word, the member list, the subset size and its 18 checkpoints. slot_of_hyp == null
 * selects the single-subset mode. */
struct ed_ball {
    const uint32_t* slot_of_hyp;
    const unsigned long long* member; /* [slots][mwords] */
    const uint32_t* wprefix;          /* [slots][mwords] */
    const uint32_t* S_of_slot;        /* [slots] */
    const uint32_t* tests_of_slot;    /* [slots][18] */
    const int32_t* list;              /* [slots][list_stride], ascending */
    const float4* centre_pos;         /* [slots] xyz of the centre */
    uint32_t mwords, list_stride;
    float radius;
    /* The scan positions the FIRST checkpoint needs (the first B positions of a slot's list), once more in the
     * scene's curve order: entry = (sorted slot of the point, its scan position). The prefix kernel walks this
     * list instead of the neighbourhood list — consecutive lanes then take neighbouring points: one coalesced
     * load instead of a gather through the index list, and voxel / model gathers that share lines (the scan order,
     * ascending scene index, is unrelated to space). [slots][plist_stride], plist_n[slot] entries; null = unused */
    const uint2* plist;
    const float4* plist_p; /* (x, y, z, sorted slot) of the same entries */
    /* every member of the neighbourhood the same way (entry = sorted slot, scan position; unmasked points only, a wave's
     * 64 consecutive scene points together): what the continuation streams instead of looking up the rank of every
     * scene point of a tile in the membership words (two gathers over addresses unrelated to space).
     * [slots][blist_stride], blist_n[slot] entries; null = unused */
    const uint2* blist;
    const uint32_t* blist_n;
    uint32_t blist_stride;
    const uint32_t* plist_n;
    uint32_t plist_stride;
};

/* builds ed_ball::plist: one thread per (sorted scene slot, neighbourhood); wave-aggregated append, so that the 64
 * consecutive points of a wave stay together in the list */
__global__ __launch_bounds__(1024) void k_ball_prefix_list(const float4* __restrict__ spos, uint32_t n_sorted,
                                                           const unsigned long long* __restrict__ member,
                                                           const uint32_t* __restrict__ wprefix, uint32_t mwords,
                                                           const uint32_t* __restrict__ B_of_slot, uint32_t stride,
                                                           uint32_t* __restrict__ plist_n, uint2* __restrict__ plist,
                                                           float4* __restrict__ plist_p, uint32_t* __restrict__ blist_n,
                                                           uint2* __restrict__ blist, uint32_t blist_stride,
                                                           const float4* __restrict__ tiles, uint32_t n_tiles,
                                                           const float4* __restrict__ centre_pos, float radius,
                                                           float ball_margin) {
    /* one list cursor per neighbourhood: an atomic per wave on it (3,200 per neighbourhood, all neighbourhoods' cursors
     * in one cache line) serialised to 0.26 ms; a workgroup of 16 waves reserves its entries with one */
    __shared__ uint32_t s_na[16], s_np[16], s_base[2];
    const uint32_t sidx = blockIdx.x * 1024 + threadIdx.x, c = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool in = false, member_pt = false;
    uint32_t pos = 0;
    float4 P = make_float4(0.f, 0.f, 0.f, 0.f);
    /* a wave = one 64-point tile: most tiles are out of the neighbourhood's reach (the same conservative test as the
     * scoring kernels'), and their points need no membership look-up */
    const uint32_t t = sidx >> 6;
    bool reach = t < n_tiles;
    if (reach) {
        const float4 tb = tiles[t], C = centre_pos[c];
        reach = !(tb.w < 0.f) && !(len(make3(tb.x, tb.y, tb.z) - make3(C.x, C.y, C.z)) > radius + tb.w * 1.00001f + ball_margin);
    }
    if (reach && sidx < n_sorted) {
        P = spos[sidx];
        if (P.x == P.x) { /* masked / non-finite / padding points carry NaN: they have no bits (reference :434) */
            const uint32_t orig = __float_as_uint(P.w);
            const unsigned long long mbits = member[(size_t)c * mwords + (orig >> 6)];
            if ((mbits >> (orig & 63u)) & 1ull) {
                pos = wprefix[(size_t)c * mwords + (orig >> 6)] + (uint32_t)__popcll(mbits & ((1ull << (orig & 63u)) - 1ull));
                member_pt = true;
                in = pos < B_of_slot[c];
            }
        }
    }
    const unsigned long long ba = __ballot(member_pt && blist != nullptr), b = __ballot(in);
    if (lane == 0) {
        s_na[wave] = (uint32_t)__popcll(ba);
        s_np[wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    if (threadIdx.x < 2u) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < 16u; ++w) tot += threadIdx.x == 0 ? s_na[w] : s_np[w];
        s_base[threadIdx.x] = tot ? atomicAdd(threadIdx.x == 0 ? &blist_n[c] : &plist_n[c], tot) : 0u;
    }
    __syncthreads();
    uint32_t base_a = s_base[0], base = s_base[1];
    for (uint32_t w = 0; w < wave; ++w) {
        base_a += s_na[w];
        base += s_np[w];
    }
    if (member_pt && blist) { /* ed_ball::blist: every member, same order */
        const uint32_t e = base_a + (uint32_t)__popcll(ba & ((1ull << lane) - 1ull));
        if (e < blist_stride) blist[(size_t)c * blist_stride + e] = make_uint2(sidx, pos);
    }
    if (in) {
        const uint32_t e = base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (e < stride) {
            plist[(size_t)c * stride + e] = make_uint2(sidx, pos);
            plist_p[(size_t)c * stride + e] = make_float4(P.x, P.y, P.z, __uint_as_float(sidx));
        }
    }
}

struct ed_args {
    ed_ball ball;
    score_args sc;           /* sref = scene oref (original order) */
    const float4* opos;      /* original order positions */
    const float4* otan;      /* .w = mask */
    const int32_t* subset;   /* scan order, or null = identity */
    uint32_t S;
    uint32_t words;          /* ceil(S / 64) */
    uint32_t h0;             /* first hypothesis of this chunk */
    unsigned long long* tail_bits;
    unsigned long long* corr_bits;
    uint32_t prefix_words;   /* k_ed_fused<true>: 64-bit words of scan positions its LDS bit arrays hold */
    const uint32_t* stop;    /* tm_search_batch: index of the first sufficient hypothesis found so far, 0xFFFFFFFF =
                                none yet; the kernels of later segments leave at once when it is set. null = unused */
    /* One segment step of SEVERAL independent batches in one launch (tm_search_batches: the chunks of a round):
     * the prefix kernel's workgroup (x, y) takes hypothesis rng_starts[y] + seg_step * seg_len + x of batch y and
     * leaves at once when that is past the batch's end or the batch's stop word rng_stop[y] is set; h0 = 0 and the
     * pending list holds global hypothesis indices. null = one batch per launch (h0, stop). */
    const uint32_t* rng_starts;
    const uint32_t* rng_stop;
    uint32_t seg_len, seg_step;
    /* k_ed_prefix_group in a search (only the FIRST SUFFICIENT hypothesis of a batch is reported, and with sufficient >=
     * the rule's threshold that is never a dropped one): 1 = a hypothesis whose count of NEAR points (brick code: an upper bound of its correspondences) already
     * fails the first checkpoint is dropped without its voxel / model gathers */
    uint32_t bound_first = 0;
    /* The continuation in waves (k_pending_sort): the pending rows are in visit order, row_info[row] = position of
     * the row among its batch's pending rows | batch << 20; a launch takes the rows with ord_lo <= position < ord_hi
     * whose batch has no sufficient hypothesis yet. Only the FIRST sufficient hypothesis of a batch counts
     * (reference include/impl/scene.hpp:330-338), and a segment with a true match holds dozens of pending
     * hypotheses that all scan their whole neighbourhood: the first few in visit order settle it. null = all rows. */
    const uint32_t* row_info;
    uint32_t ord_lo, ord_hi;
};

/* does this launch take pending row `row` (see ed_args::row_info)? */
__device__ __forceinline__ bool ed_row_selected(const ed_args& a, uint32_t row) {
    if (!a.row_info) return true;
    const uint32_t info = a.row_info[row], ord = info & 0xFFFFFu, b = info >> 20;
    if (ord < a.ord_lo || ord >= a.ord_hi) return false;
    if (a.rng_stop) return a.rng_stop[b] == 0xFFFFFFFFu;
    return !(a.stop && *a.stop != 0xFFFFFFFFu);
}

__device__ __forceinline__ bool ed_stopped(const ed_args& a) { return a.stop && *a.stop != 0xFFFFFFFFu; }

__global__ __launch_bounds__(256) void k_ed_bits(ed_args a) {
    const uint32_t pos = blockIdx.x * 256 + threadIdx.x;
    const uint32_t hl = blockIdx.y;
    bool tail = false;
    uint32_t corr = 0;
    if (pos < a.S) {
        uint32_t idx = a.subset ? (uint32_t)a.subset[pos] : pos;
        float4 P = a.opos[idx];
        if (a.otan[idx].w == 0.f) { /* mask_, reference :434 */
            const pose12 T = load_pose(a.sc.poses + (size_t)(a.h0 + hl) * 16);
            long long fx = 0;
            uint32_t pass = 0;
            corr = eval_point(a.sc, T, make3(P.x, P.y, P.z), idx, fx, pass, nullptr, &tail);
        }
    }
    unsigned long long tb = __ballot(tail);
    unsigned long long cb = __ballot(corr != 0);
    if ((threadIdx.x & 63) == 0 && pos < a.S) {
        a.tail_bits[(size_t)hl * a.words + pos / 64] = tb;
        a.corr_bits[(size_t)hl * a.words + pos / 64] = cb;
    }
}

/* position of every scene point in the scan order (0xFFFFFFFF = not in the subset) */
__global__ __launch_bounds__(256) void k_ed_positions(const int32_t* __restrict__ subset, uint32_t S,
                                                      uint32_t* __restrict__ pos_of_orig) {
    const uint32_t pos = blockIdx.x * 256 + threadIdx.x;
    if (pos < S) pos_of_orig[subset ? (uint32_t)subset[pos] : pos] = pos;
}
/* a point named twice keeps only one of its positions: flag it */
__global__ __launch_bounds__(256) void k_ed_check_positions(const int32_t* __restrict__ subset, uint32_t S,
                                                            const uint32_t* __restrict__ pos_of_orig,
                                                            uint32_t* __restrict__ duplicate) {
    const uint32_t pos = blockIdx.x * 256 + threadIdx.x;
    if (pos < S && pos_of_orig[(uint32_t)subset[pos]] != pos) *duplicate = 1u;
}

/* which tiles / super-tiles hold at least one live point of the subset (bit per tile in 64-bit words:
 * word u = super-tile u) */
__global__ __launch_bounds__(256) void k_ed_tile_active(const float4* __restrict__ spos,
                                                        const uint32_t* __restrict__ pos_of_orig, uint32_t n_tiles,
                                                        unsigned long long* __restrict__ tile_bits) {
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const float4 P = spos[(size_t)t * 64 + (threadIdx.x & 63)];
    const bool in = (P.x == P.x) && pos_of_orig[__float_as_uint(P.w)] != 0xFFFFFFFFu;
    if (__ballot(in) && (threadIdx.x & 63) == 0) atomicOr(&tile_bits[t >> 6], 1ull << (t & 63u));
}

/* membership bits of several balls at once: blockIdx.y = slot */
/* a thread keeps its point and walks the slots (the centre is uniform: scalar loads): the cloud is read once per call
 * instead of once per slot (20 slots x 16 MB per search step: 67 us of L2 traffic for 2e7 distance tests) */
__global__ __launch_bounds__(256) void k_ball_members(const float4* __restrict__ opos, uint32_t n,
                                                      const uint32_t* __restrict__ centres, uint32_t n_slots, float r2,
                                                      uint32_t mwords, unsigned long long* __restrict__ member,
                                                      float4* __restrict__ centre_pos) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const float4 p = i < n ? opos[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t c = 0; c < n_slots; ++c) {
        const float4 C = opos[centres[c]];
        const bool in = i < n && len2(make3(p.x, p.y, p.z) - make3(C.x, C.y, C.z)) <= r2;
        const unsigned long long b = __ballot(in);
        if ((threadIdx.x & 63) == 0 && (i >> 6) < mwords) member[(size_t)c * mwords + (i >> 6)] = b;
        if (i == 0) centre_pos[c] = C;
    }
}

/* The tangent members of several balls, in ascending index order: one 1024-thread workgroup per slot walks the scene's
 * tangent points (ascending), keeps those inside the ball, and writes (rank of the point in the ball's ascending index
 * order, index) at the position a block scan gives it — staging[slot][e], t_counts[slot] entries. */
__global__ __launch_bounds__(1024) void k_ball_tangent(const unsigned long long* __restrict__ member,
                                                       const uint32_t* __restrict__ wprefix, uint32_t mwords,
                                                       const uint32_t* __restrict__ tangent_idx, uint32_t n_tangent,
                                                       uint32_t stride, uint2* __restrict__ staging,
                                                       uint32_t* __restrict__ t_counts) {
    __shared__ uint32_t s_sum[1024];
    const uint32_t c = blockIdx.x, t = threadIdx.x;
    const unsigned long long* mb = member + (size_t)c * mwords;
    const uint32_t per = (n_tangent + 1023u) / 1024u, k0 = min(t * per, n_tangent), k1 = min(k0 + per, n_tangent);
    uint32_t local = 0;
    for (uint32_t k = k0; k < k1; ++k) {
        const uint32_t p = tangent_idx[k];
        local += (uint32_t)((mb[p >> 6] >> (p & 63u)) & 1ull);
    }
    s_sum[t] = local;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) { /* Hillis-Steele inclusive scan */
        const uint32_t v = t >= off ? s_sum[t - off] : 0u;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    uint32_t e = s_sum[t] - local;
    for (uint32_t k = k0; k < k1; ++k) {
        const uint32_t p = tangent_idx[k];
        const unsigned long long mbits = mb[p >> 6];
        if (!((mbits >> (p & 63u)) & 1ull)) continue;
        const uint32_t rank = wprefix[(size_t)c * mwords + (p >> 6)] + (uint32_t)__popcll(mbits & ((1ull << (p & 63u)) - 1ull));
        if (e < stride) staging[(size_t)c * stride + e] = make_uint2(rank, p);
        ++e;
    }
    if (t == 1023) t_counts[c] = s_sum[1023];
}

/* the slots' lists one behind the other (CSR: slot c starts at the sum of the counts before it) */
__global__ __launch_bounds__(256) void k_ball_tangent_compact(const uint2* __restrict__ staging, uint32_t stride,
                                                              const uint32_t* __restrict__ t_counts,
                                                              uint2* __restrict__ out) {
    const uint32_t c = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= t_counts[c]) return;
    uint32_t base = 0;
    for (uint32_t q = 0; q < c; ++q) base += t_counts[q];
    out[base + e] = staging[(size_t)c * stride + e];
}

/* exclusive prefix of the word popcounts of one slot (one 1024-thread block per slot) and the total */
__global__ __launch_bounds__(1024) void k_ball_prefix(const unsigned long long* __restrict__ member, uint32_t mwords,
                                                      uint32_t* __restrict__ wprefix, uint32_t* __restrict__ S_of_slot) {
    __shared__ uint32_t s_sum[1024];
    const uint32_t c = blockIdx.x, t = threadIdx.x;
    const unsigned long long* mb = member + (size_t)c * mwords;
    const uint32_t per = (mwords + 1023) / 1024, w0 = t * per, w1 = min(w0 + per, mwords);
    uint32_t local = 0;
    for (uint32_t w = w0; w < w1; ++w) local += (uint32_t)__popcll(mb[w]);
    s_sum[t] = local;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) { /* Hillis-Steele inclusive scan */
        uint32_t v = t >= off ? s_sum[t - off] : 0u;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    uint32_t run = s_sum[t] - local;
    for (uint32_t w = w0; w < w1; ++w) {
        wprefix[(size_t)c * mwords + w] = run;
        run += (uint32_t)__popcll(mb[w]);
    }
    if (t == 1023) S_of_slot[c] = s_sum[1023];
}

/* member lists (ascending original index): blockIdx.y = slot */
__global__ __launch_bounds__(256) void k_ball_list(const unsigned long long* __restrict__ member,
                                                   const uint32_t* __restrict__ wprefix, uint32_t n, uint32_t mwords,
                                                   uint32_t list_stride, int32_t* __restrict__ list) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (i >= n) return;
    const unsigned long long b = member[(size_t)c * mwords + (i >> 6)];
    if (!((b >> (i & 63u)) & 1ull)) return;
    const uint32_t pos = wprefix[(size_t)c * mwords + (i >> 6)] + (uint32_t)__popcll(b & ((1ull << (i & 63u)) - 1ull));
    list[(size_t)c * list_stride + pos] = (int32_t)i;
}

struct ed_cull {
    const unsigned long long* tile_bits; /* per super-tile: which of its 64 tiles hold subset points */
    const float4* spos;   /* Morton order, w = bits of the original index, x = NaN for masked / non-finite */
    const float4* tiles;  /* sphere per 64 sorted points */
    const float4* supers; /* sphere per 4096 sorted points */
    const uint32_t* pos_of_orig;
    uint32_t n_tiles;
    float coord_max;
};

constexpr int kEdChunk = 4096; /* tiles per culling round = 64 super-tiles */

__global__ __launch_bounds__(256) void k_ed_bits_culled(ed_args a, ed_cull ca) {
    __shared__ uint32_t s_list[kEdChunk];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_smask;
    const uint32_t hl = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const pose12 T = load_pose(a.sc.poses + (size_t)(a.h0 + hl) * 16);
    const float rnorm = sqrtf(((T.r00 * T.r00 + T.r10 * T.r10) + (T.r20 * T.r20 + T.r01 * T.r01)) +
                              ((T.r11 * T.r11 + T.r21 * T.r21) + (T.r02 * T.r02 + T.r12 * T.r12)) +
                              T.r22 * T.r22);
    const float abs_eps = 1e-6f * (3.f * rnorm * ca.coord_max + fabsf(T.tx) + fabsf(T.ty) + fabsf(T.tz) + 1.f);
    unsigned long long* tb = a.tail_bits + (size_t)hl * a.words;
    unsigned long long* cb = a.corr_bits + (size_t)hl * a.words;
    const bool ball = a.ball.slot_of_hyp != nullptr;
    const uint32_t slot = ball ? a.ball.slot_of_hyp[a.h0 + hl] : 0u;
    f3 bc = make3(0.f, 0.f, 0.f);
    if (ball) {
        float4 C = a.ball.centre_pos[slot];
        bc = make3(C.x, C.y, C.z);
    }
    const float ball_margin = 1e-5f * a.ball.radius + 4e-6f * ca.coord_max + 1e-12f;
    const unsigned long long* member = ball ? a.ball.member + (size_t)slot * a.ball.mwords : nullptr;
    const uint32_t* wprefix = ball ? a.ball.wprefix + (size_t)slot * a.ball.mwords : nullptr;
    for (uint32_t t0 = 0; t0 < ca.n_tiles; t0 += kEdChunk) {
        if (wave == 0) {
            const uint32_t u = (t0 >> 6) + lane;
            bool hit = false;
            if (u * 64u < ca.n_tiles && (ball || ca.tile_bits[u])) {
                float4 sb = ca.supers[u];
                hit = sphere_hits_box(a.sc, T, rnorm, abs_eps, sb);
                if (ball) hit = hit && !(len(make3(sb.x, sb.y, sb.z) - bc) > a.ball.radius + sb.w * 1.00001f + ball_margin);
            }
            unsigned long long mask = __ballot(hit);
            if (lane == 0) {
                s_smask = mask;
                s_n = 0;
            }
        }
        __syncthreads();
        const unsigned long long smask = s_smask;
        for (uint32_t r = wave; r < (uint32_t)(kEdChunk / 64); r += 4) {
            if (!((smask >> r) & 1ull)) continue;
            const uint32_t t = t0 + r * 64u + lane;
            if (t < ca.n_tiles && (ball || ((ca.tile_bits[t >> 6] >> lane) & 1ull))) {
                float4 tbs = ca.tiles[t];
                bool hit = sphere_hits_box(a.sc, T, rnorm, abs_eps, tbs);
                if (ball) hit = hit && !(len(make3(tbs.x, tbs.y, tbs.z) - bc) > a.ball.radius + tbs.w * 1.00001f + ball_margin);
                if (hit) s_list[atomicAdd(&s_n, 1u)] = t;
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        for (uint32_t i = wave; i < n; i += 4) {
            const uint32_t sidx = s_list[i] * 64u + lane;
            const float4 P = ca.spos[sidx];
            if (!(P.x == P.x)) continue; /* mask_ (:434), non-finite, padding */
            uint32_t pos;
            if (ball) { /* position = rank of the point inside its centre's neighbourhood */
                const uint32_t orig = __float_as_uint(P.w);
                const unsigned long long mbits = member[orig >> 6];
                if (!((mbits >> (orig & 63u)) & 1ull)) continue;
                pos = wprefix[orig >> 6] + (uint32_t)__popcll(mbits & ((1ull << (orig & 63u)) - 1ull));
            } else {
                pos = ca.pos_of_orig[__float_as_uint(P.w)];
                if (pos == 0xFFFFFFFFu) continue;
            }
            long long fx = 0;
            uint32_t pass = 0;
            bool tail = false;
            const uint32_t corr = eval_point(a.sc, T, make3(P.x, P.y, P.z), sidx, fx, pass, nullptr, &tail);
            if (tail) {
                const unsigned long long bit = 1ull << (pos & 63u);
                atomicOr(&tb[pos >> 6], bit);
                if (corr) atomicOr(&cb[pos >> 6], bit);
            }
        }
        __syncthreads();
    }
}

struct ed_out {
    uint32_t* counts;
    int32_t* dropped_at;
    uint32_t* saved;
    uint32_t* limit; /* last position whose score term counts, or 0xFFFFFFFF for none */
};

__global__ __launch_bounds__(64) void k_ed_walk(ed_args a, ed_out o, float accept_thresh,
                                                const uint32_t* __restrict__ tests,
                                                const uint32_t* __restrict__ pending = nullptr,
                                                const uint32_t* __restrict__ n_pending = nullptr, uint32_t row0 = 0) {
    /* with a pending list: bit row blockIdx.x belongs to hypothesis pending[row0 + blockIdx.x] */
    const uint32_t row = blockIdx.x, lane = threadIdx.x;
    if (ed_stopped(a)) return;
    if (pending && row0 + row >= *n_pending) return;
    if (pending && !ed_row_selected(a, row0 + row)) return;
    const uint32_t hl = pending ? pending[row0 + row] : row;
    const bool ball = a.ball.slot_of_hyp != nullptr;
    const uint32_t slot = ball ? a.ball.slot_of_hyp[a.h0 + hl] : 0u;
    const uint32_t S = ball ? a.ball.S_of_slot[slot] : a.S;
    if (ball) tests = a.ball.tests_of_slot + 18u * slot;
    const unsigned long long* tb = a.tail_bits + (size_t)row * a.words;
    const unsigned long long* cb = a.corr_bits + (size_t)row * a.words;
    uint32_t k = 0;           /* next_test */
    uint32_t min_pos = 0;     /* earliest 0-based position the next test may run at */
    if (tests[0] > 0) min_pos = tests[0] - 1;
    uint32_t corr_before = 0; /* correspondences in the words before this chunk */
    int32_t dropped = -1;
    uint32_t drop_count = 0, drop_pos = 0;
    for (uint32_t w0 = 0; w0 < a.words && dropped < 0; w0 += 64) {
        const uint32_t w = w0 + lane;
        const unsigned long long tword = w < a.words ? tb[w] : 0ull;
        const unsigned long long cword = w < a.words ? cb[w] : 0ull;
        /* inclusive prefix of popcounts over lanes */
        uint32_t pc = (uint32_t)__popcll(cword);
        uint32_t incl = pc;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t v = __shfl_up(incl, off);
            if ((int)lane >= off) incl += v;
        }
        const uint32_t chunk_end = (w0 + 64) * 64; /* exclusive, in positions */
        while (k < 18 && min_pos < chunk_end) {
            unsigned long long bits = tword;
            const uint32_t wbase = w * 64;
            if (min_pos >= wbase + 64)
                bits = 0ull;
            else if (min_pos > wbase)
                bits &= ~0ull << (min_pos - wbase);
            unsigned long long have = __ballot(bits != 0ull);
            if (!have) break; /* no test point left in this chunk */
            const int L = __ffsll((long long)have) - 1;
            const unsigned long long lbits = __shfl(bits, L);
            const unsigned long long lcorr = __shfl(cword, L);
            const uint32_t lincl = __shfl(incl, L);
            const uint32_t lpc = __shfl(pc, L);
            const int b = __ffsll((long long)lbits) - 1;
            const uint32_t pos = (w0 + (uint32_t)L) * 64 + (uint32_t)b;
            const unsigned long long upto = (b == 63) ? ~0ull : ((2ull << b) - 1ull);
            const uint32_t corrs = corr_before + (lincl - lpc) + (uint32_t)__popcll(lcorr & upto);
            const uint32_t tried = pos + 1;
            const uint32_t upper = early_drop_upper(tried, S, corrs);
            if ((float)upper < accept_thresh) { /* reference :500 */
                dropped = (int32_t)k;
                drop_count = corrs;
                drop_pos = pos;
                break;
            }
            ++k;
            min_pos = pos + 1;
            if (k < 18 && tests[k] > 0 && tests[k] - 1 > min_pos) min_pos = tests[k] - 1;
        }
        corr_before += __shfl(incl, 63);
    }
    if (lane == 0) {
        if (dropped >= 0) {
            o.counts[a.h0 + hl] = drop_count;
            o.dropped_at[a.h0 + hl] = dropped;
            o.saved[a.h0 + hl] = S - (drop_pos + 1);
            o.limit[a.h0 + hl] = drop_pos;
        } else {
            o.counts[a.h0 + hl] = corr_before;
            o.dropped_at[a.h0 + hl] = -1;
            o.saved[a.h0 + hl] = 0;
            o.limit[a.h0 + hl] = S ? S - 1 : 0xFFFFFFFFu;
        }
    }
}

__global__ __launch_bounds__(256) void k_ed_score(ed_args a, const uint32_t* __restrict__ limit) {
    const uint32_t pos = blockIdx.x * 256 + threadIdx.x;
    const uint32_t hl = blockIdx.y;
    long long fx = 0;
    const bool ball = a.ball.slot_of_hyp != nullptr;
    const uint32_t slot = ball ? a.ball.slot_of_hyp[a.h0 + hl] : 0u;
    const uint32_t S = ball ? a.ball.S_of_slot[slot] : a.S;
    if (pos < S) {
        const uint32_t lim = limit[a.h0 + hl];
        const unsigned long long cw = a.corr_bits[(size_t)hl * a.words + pos / 64];
        if (lim != 0xFFFFFFFFu && pos <= lim && ((cw >> (pos & 63)) & 1ull)) {
            uint32_t idx = ball ? (uint32_t)a.ball.list[(size_t)slot * a.ball.list_stride + pos]
                                : (a.subset ? (uint32_t)a.subset[pos] : pos);
            float4 P = a.opos[idx];
            const pose12 T = load_pose(a.sc.poses + (size_t)(a.h0 + hl) * 16);
            uint32_t pass = 0;
            (void)eval_point(a.sc, T, make3(P.x, P.y, P.z), idx, fx, pass, nullptr, nullptr);
        }
    }
    for (int off = 32; off > 0; off >>= 1) fx += __shfl_down(fx, off);
    if ((threadIdx.x & 63) == 0 && fx) atomicAdd(&a.sc.scores_fx[a.h0 + hl], (unsigned long long)fx);
}

/* Wide continuation of the early-drop rule. The ~2 % of hypotheses that survive the first checkpoint are few, so
 * one workgroup each (k_ed_fused<false>) leaves most of the chip idle while each of them culls the whole scene
 * and walks its neighbourhood. Here the same three passes run over (pending hypothesis, part of the scene):
 *   k_ed_zero_rows   clears the bit rows of the pending hypotheses
 *   k_ed_bits_wide   a workgroup = one super-tile (64 tiles) of the sorted scene for one pending hypothesis, a
 *                    wave 16 of its tiles: sphere tests, then the points of the surviving tiles -> (tail,
 *                    correspondence) bits at their scan positions (global atomicOr), as k_ed_bits_culled does
 *   k_ed_walk        the rule on those rows
 *   k_ed_score_rows  score terms of the correspondences up to the stop position, eight positions per thread, one
 *                    atomic per workgroup (same-address atomics serialise: one per wave cost 0.4 ms) */
__global__ __launch_bounds__(256) void k_ed_zero_rows(unsigned long long* __restrict__ tail_bits,
                                                      unsigned long long* __restrict__ corr_bits, uint32_t words,
                                                      const uint32_t* __restrict__ n_pending, uint32_t row0,
                                                      uint32_t rows_cap) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n_all = *n_pending;
    const uint32_t n_rows = min(n_all > row0 ? n_all - row0 : 0u, rows_cap);
    if (w >= words) return;
    for (uint32_t row = blockIdx.y; row < n_rows; row += gridDim.y) {
        tail_bits[(size_t)row * words + w] = 0ull;
        corr_bits[(size_t)row * words + w] = 0ull;
    }
}

__global__ __launch_bounds__(256) void k_ed_bits_wide(ed_args a, ed_cull ca, const uint32_t* __restrict__ pending,
                                                      const uint32_t* __restrict__ n_pending, uint32_t row0,
                                                      uint32_t rows_cap) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t u = blockIdx.x; /* super-tile of this workgroup; wave w takes its tiles 16 w .. 16 w + 15 */
    if (u * 64u >= ca.n_tiles || ed_stopped(a)) return;
    const uint32_t n_all = *n_pending;
    const uint32_t n_rows = min(n_all > row0 ? n_all - row0 : 0u, rows_cap);
    const bool ball = a.ball.slot_of_hyp != nullptr;
    const float4 sb = ca.supers[u];
    const uint32_t t = u * 64u + wave * 16u + (lane & 15u);
    const bool have_tile = lane < 16u && t < ca.n_tiles && (ball || ((ca.tile_bits[u] >> (t & 63u)) & 1ull));
    const float4 tbs = have_tile ? ca.tiles[t] : make_float4(0.f, 0.f, 0.f, -1.f);
    const float ball_margin = 1e-5f * a.ball.radius + 4e-6f * ca.coord_max + 1e-12f;
    for (uint32_t row = blockIdx.y; row < n_rows; row += gridDim.y) {
        if (!ed_row_selected(a, row0 + row)) continue;
        const uint32_t hl = pending[row0 + row];
        const pose12 T = load_pose(a.sc.poses + (size_t)(a.h0 + hl) * 16);
        const float rnorm = sqrtf(((T.r00 * T.r00 + T.r10 * T.r10) + (T.r20 * T.r20 + T.r01 * T.r01)) +
                                  ((T.r11 * T.r11 + T.r21 * T.r21) + (T.r02 * T.r02 + T.r12 * T.r12)) +
                                  T.r22 * T.r22);
        const float abs_eps = 1e-6f * (3.f * rnorm * ca.coord_max + fabsf(T.tx) + fabsf(T.ty) + fabsf(T.tz) + 1.f);
        const uint32_t slot = ball ? a.ball.slot_of_hyp[a.h0 + hl] : 0u;
        f3 bc = make3(0.f, 0.f, 0.f);
        if (ball) {
            const float4 C = a.ball.centre_pos[slot];
            bc = make3(C.x, C.y, C.z);
        }
        /* the super-tile as a whole (the same test in every lane), then one tile per lane */
        bool shit = (ball || ca.tile_bits[u]) && sphere_hits_box(a.sc, T, rnorm, abs_eps, sb);
        if (ball) shit = shit && !(len(make3(sb.x, sb.y, sb.z) - bc) > a.ball.radius + sb.w * 1.00001f + ball_margin);
        if (!shit) continue;
        bool hit = have_tile && sphere_hits_box(a.sc, T, rnorm, abs_eps, tbs);
        if (ball) hit = hit && !(len(make3(tbs.x, tbs.y, tbs.z) - bc) > a.ball.radius + tbs.w * 1.00001f + ball_margin);
        unsigned long long tmask = __ballot(hit);
        const unsigned long long* member = ball ? a.ball.member + (size_t)slot * a.ball.mwords : nullptr;
        const uint32_t* wprefix = ball ? a.ball.wprefix + (size_t)slot * a.ball.mwords : nullptr;
        unsigned long long* tb = a.tail_bits + (size_t)row * a.words;
        unsigned long long* cb = a.corr_bits + (size_t)row * a.words;
        while (tmask) {
            const uint32_t tile = u * 64u + wave * 16u + (uint32_t)__builtin_ctzll(tmask);
            tmask &= tmask - 1ull;
            const uint32_t sidx = tile * 64u + lane;
            const float4 P = ca.spos[sidx];
            if (!(P.x == P.x)) continue; /* mask_ (:434), non-finite, padding */
            uint32_t pos;
            if (ball) { /* position = rank of the point inside its centre's neighbourhood */
                const uint32_t orig = __float_as_uint(P.w);
                const unsigned long long mbits = member[orig >> 6];
                if (!((mbits >> (orig & 63u)) & 1ull)) continue;
                pos = wprefix[orig >> 6] + (uint32_t)__popcll(mbits & ((1ull << (orig & 63u)) - 1ull));
            } else {
                pos = ca.pos_of_orig[__float_as_uint(P.w)];
                if (pos == 0xFFFFFFFFu) continue;
            }
            long long fx = 0;
            uint32_t pass = 0;
            bool tail = false;
            const uint32_t corr = eval_point(a.sc, T, make3(P.x, P.y, P.z), sidx, fx, pass, nullptr, &tail);
            if (tail) {
                const unsigned long long bit = 1ull << (pos & 63u);
                atomicOr(&tb[pos >> 6], bit);
                if (corr) atomicOr(&cb[pos >> 6], bit);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_ed_score_rows(ed_args a, const uint32_t* __restrict__ limit,
                                                       const uint32_t* __restrict__ pending,
                                                       const uint32_t* __restrict__ n_pending, uint32_t row0,
                                                       uint32_t rows_cap) {
    __shared__ unsigned long long s_part[4];
    if (ed_stopped(a)) return;
    const uint32_t byte = blockIdx.x * 256 + threadIdx.x, w = byte >> 3; /* eight scan positions per thread */
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n_all = *n_pending;
    const uint32_t n_rows = min(n_all > row0 ? n_all - row0 : 0u, rows_cap);
    const bool ball = a.ball.slot_of_hyp != nullptr;
    for (uint32_t row = blockIdx.y; row < n_rows; row += gridDim.y) {
        if (!ed_row_selected(a, row0 + row)) continue; /* uniform over the workgroup */
        const uint32_t hl = pending[row0 + row];
        const uint32_t lim = limit[a.h0 + hl];
        long long fx = 0;
        if (lim != 0xFFFFFFFFu && w <= (lim >> 6) && w < a.words) {
            unsigned long long bitsw = a.corr_bits[(size_t)row * a.words + w];
            if (w == (lim >> 6) && (lim & 63u) != 63u) bitsw &= (2ull << (lim & 63u)) - 1ull;
            uint32_t bits8 = (uint32_t)(bitsw >> (8u * (byte & 7u))) & 0xFFu;
            if (bits8) {
                const uint32_t slot = ball ? a.ball.slot_of_hyp[a.h0 + hl] : 0u;
                const pose12 T = load_pose(a.sc.poses + (size_t)(a.h0 + hl) * 16);
                while (bits8) {
                    const uint32_t pos = byte * 8u + (uint32_t)__builtin_ctz(bits8);
                    bits8 &= bits8 - 1u;
                    const uint32_t idx = ball ? (uint32_t)a.ball.list[(size_t)slot * a.ball.list_stride + pos]
                                              : (a.subset ? (uint32_t)a.subset[pos] : pos);
                    const float4 P = a.opos[idx];
                    uint32_t pass = 0;
                    (void)eval_point(a.sc, T, make3(P.x, P.y, P.z), idx, fx, pass, nullptr, nullptr);
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) fx += __shfl_down(fx, off);
        if (lane == 0) s_part[wave] = (unsigned long long)fx;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long sum = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
            if (sum) atomicAdd(&a.sc.scores_fx[a.h0 + hl], sum);
        }
        __syncthreads();
    }
}

/* Wide continuation without bit rows. The rule needs, per checkpoint k, (1) the first evaluated ("tail") scan
 * position at or after x_k = max(tests[k], 1) - 1 and (2) the correspondences up to and including it. As long as
 * every interval [x_k, x_k+1) holds a tail position — any hypothesis that is still alive after the first
 * checkpoint has thousands — test k runs at m_k = min{tail pos >= x_k}, which lies inside that interval, and the
 * positions between x_k and m_k have no bits at all, so
 *     corrs_k = #{correspondences at pos < x_k} + [m_k is a correspondence].
 * Both are order-free reductions over the neighbourhood: per row 19 interval counts, 19 interval sums of the
 * fixed-point score terms, and 18 packed minima (position << 42 | not-a-correspondence << 41 | score term), kept in
 * LDS per (row, super-tile) workgroup and merged with one global atomic per non-empty entry. k_ed_hist_walk then
 * applies the rule to the 18 minima; a row with an empty interval before a later tail position (the next test
 * position then depends on bit order), or with a score term that does not fit 41 bits, goes to the bit-array
 * kernel through a fallback list — same outcome either way (tests: every option combination against the oracle).
 * Layout of a row (kEdHistWords 64-bit words): [0, 10) 20 x u32 counts (19 intervals + overflow flag),
 * [10, 29) sums, [29, 47) minima. */
constexpr uint32_t kEdHistWords = 48, kEdHistFx = 10, kEdHistMin = 29;
constexpr uint32_t kEdHistPosBits = 22; /* scan positions below 4,194,304 */
#ifndef TM_ED_NT
#define TM_ED_NT 2
#endif
constexpr int kEdNT = TM_ED_NT;                /* evaluations a lane keeps in flight (eval_points_staged) */

__global__ __launch_bounds__(256) void k_ed_hist_init(unsigned long long* __restrict__ hist,
                                                      const uint32_t* __restrict__ n_pending, uint32_t row0,
                                                      uint32_t rows_cap, uint32_t* __restrict__ fb_n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2u) fb_n[i] = 0u;
    const uint32_t n_all = *n_pending;
    const uint32_t n_rows = min(n_all > row0 ? n_all - row0 : 0u, rows_cap);
    if (i < n_rows * kEdHistWords) hist[i] = (i % kEdHistWords) >= kEdHistMin ? ~0ull : 0ull;
}

__global__ __launch_bounds__(256) void k_ed_hist_wide(ed_args a, ed_cull ca, const uint32_t* __restrict__ pending,
                                                      const uint32_t* __restrict__ n_pending, uint32_t row0,
                                                      uint32_t rows_cap, unsigned long long* __restrict__ hist) {
    __shared__ unsigned long long s_h[kEdHistWords];
    __shared__ uint32_t s_x[18];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t u = blockIdx.x; /* super-tile of this workgroup; wave w takes its tiles 16 w .. 16 w + 15 */
    if (u * 64u >= ca.n_tiles || ed_stopped(a)) return;
    const uint32_t n_all = *n_pending;
    const uint32_t n_rows = min(n_all > row0 ? n_all - row0 : 0u, rows_cap);
    const float4 sb = ca.supers[u];
    const uint32_t t = u * 64u + wave * 16u + (lane & 15u);
    const bool have_tile = lane < 16u && t < ca.n_tiles;
    const float4 tbs = have_tile ? ca.tiles[t] : make_float4(0.f, 0.f, 0.f, -1.f);
    const float ball_margin = 1e-5f * a.ball.radius + 4e-6f * ca.coord_max + 1e-12f;
    uint32_t* const s_cnt = reinterpret_cast<uint32_t*>(s_h);
    for (uint32_t row = blockIdx.y; row < n_rows; row += gridDim.y) {
        if (!ed_row_selected(a, row0 + row)) continue;
        const uint32_t hl = pending[row0 + row];
        const pose12 T = load_pose(a.sc.poses + (size_t)(a.h0 + hl) * 16);
        const float rnorm = sqrtf(((T.r00 * T.r00 + T.r10 * T.r10) + (T.r20 * T.r20 + T.r01 * T.r01)) +
                                  ((T.r11 * T.r11 + T.r21 * T.r21) + (T.r02 * T.r02 + T.r12 * T.r12)) +
                                  T.r22 * T.r22);
        const float abs_eps = 1e-6f * (3.f * rnorm * ca.coord_max + fabsf(T.tx) + fabsf(T.ty) + fabsf(T.tz) + 1.f);
        const uint32_t slot = a.ball.slot_of_hyp[a.h0 + hl];
        const float4 C = a.ball.centre_pos[slot];
        const f3 bc = make3(C.x, C.y, C.z);
        /* the super-tile as a whole (the same test in every lane), then one tile per lane */
        bool shit = sphere_hits_box(a.sc, T, rnorm, abs_eps, sb);
        shit = shit && !(len(make3(sb.x, sb.y, sb.z) - bc) > a.ball.radius + sb.w * 1.00001f + ball_margin);
        if (!shit) continue;
        bool hit = have_tile && sphere_hits_box(a.sc, T, rnorm, abs_eps, tbs);
        hit = hit && !(len(make3(tbs.x, tbs.y, tbs.z) - bc) > a.ball.radius + tbs.w * 1.00001f + ball_margin);
        unsigned long long tmask = __ballot(hit);
        const uint32_t S = a.ball.S_of_slot[slot];
        const uint32_t* tests = a.ball.tests_of_slot + 18u * slot;
        if (threadIdx.x < kEdHistWords) s_h[threadIdx.x] = threadIdx.x >= kEdHistMin ? ~0ull : 0ull;
        if (threadIdx.x >= 64u && threadIdx.x < 82u) {
            const uint32_t tk = tests[threadIdx.x - 64u];
            s_x[threadIdx.x - 64u] = tk > 0u ? tk - 1u : 0u;
        }
        __syncthreads();
        const float per_pos = S ? 20.f / (float)S : 0.f;
        const unsigned long long* member = a.ball.member + (size_t)slot * a.ball.mwords;
        const uint32_t* wprefix = a.ball.wprefix + (size_t)slot * a.ball.mwords;
        while (tmask) { /* kEdNT tiles per turn, their evaluations level by level (eval_points_staged) */
            uint32_t sidx[kEdNT], pos[kEdNT];
            f3 pt[kEdNT];
            bool live[kEdNT], tail[kEdNT], corr[kEdNT];
            long long fx[kEdNT];
            float4 P[kEdNT];
#pragma unroll
            for (int i = 0; i < kEdNT; ++i) {
                live[i] = tmask != 0ull;
                const uint32_t tile = u * 64u + wave * 16u + (live[i] ? (uint32_t)__builtin_ctzll(tmask) : 0u);
                tmask &= tmask - 1ull;
                sidx[i] = tile * 64u + lane;
                P[i] = live[i] ? ca.spos[sidx[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            unsigned long long mbits[kEdNT];
            uint32_t wp[kEdNT];
#pragma unroll
            for (int i = 0; i < kEdNT; ++i) {
                live[i] = live[i] && (P[i].x == P[i].x); /* mask_ (:434), non-finite, padding */
                const uint32_t orig = __float_as_uint(P[i].w);
                mbits[i] = live[i] ? member[orig >> 6] : 0ull;
                wp[i] = live[i] ? wprefix[orig >> 6] : 0u;
            }
#pragma unroll
            for (int i = 0; i < kEdNT; ++i) {
                const uint32_t orig = __float_as_uint(P[i].w);
                live[i] = live[i] && ((mbits[i] >> (orig & 63u)) & 1ull);
                pos[i] = wp[i] + (uint32_t)__popcll(mbits[i] & ((1ull << (orig & 63u)) - 1ull));
                pt[i] = make3(P[i].x, P[i].y, P[i].z);
            }
            eval_points_staged<kEdNT>(a.sc, [&](int) -> const pose12& { return T; }, pt, sidx, live, tail, corr, fx);
#pragma unroll
            for (int i = 0; i < kEdNT; ++i) {
                if (!tail[i]) continue;
                /* interval of pos: j = #{k : x_k <= pos}; the guess is exact up to the rounding of tests[] */
                uint32_t j = min(18u, (uint32_t)((float)(pos[i] + 1u) * per_pos));
                while (j < 18u && s_x[j] <= pos[i]) ++j;
                while (j > 0u && s_x[j - 1u] > pos[i]) --j;
                if (corr[i]) {
                    atomicAdd(&s_cnt[j], 1u);
                    atomicAdd(&s_h[kEdHistFx + j], (unsigned long long)fx[i]);
                    if ((unsigned long long)fx[i] >> 41) atomicAdd(&s_cnt[19], 1u);
                }
                if (j > 0u) {
                    const unsigned long long key = ((unsigned long long)pos[i] << 42) | (corr[i] ? 0ull : 1ull << 41) |
                                                   (corr[i] ? ((unsigned long long)fx[i] & ((1ull << 41) - 1ull)) : 0ull);
                    if (key < s_h[kEdHistMin + j - 1u]) atomicMin(&s_h[kEdHistMin + j - 1u], key);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < kEdHistWords - 1u) {
            const unsigned long long v = s_h[threadIdx.x];
            unsigned long long* g = hist + (size_t)row * kEdHistWords + threadIdx.x;
            if (threadIdx.x < kEdHistMin) {
                if (v) atomicAdd(g, v);
            } else if (v != ~0ull)
                atomicMin(g, v);
        }
        __syncthreads();
    }
}

/* The same reductions from the neighbourhood's own list in curve order (ed_ball::blist) instead of the scene's
 * tiles: workgroup (c, y) takes entries [c kEdListChunk, (c + 1) kEdListChunk) (512: two per lane) of the list of every
 * row it visits — short pieces: with 2,048 or more entries per workgroup and row the batch was a fifth slower. No sphere tests
 * (the rows are hypotheses that passed the first checkpoint: most of their neighbourhood lands in the grid anyway)
 * and no rank look-ups — the position rides in the entry. A point inside the grid whose super-brick has no brick in
 * reach of the surface is a test position at most (score_args::super_near). */
#ifndef TM_ED_LIST_ROUNDS
#define TM_ED_LIST_ROUNDS 1
#endif
constexpr uint32_t kEdListRounds = TM_ED_LIST_ROUNDS; /* rounds of kEdNT entries per thread a workgroup takes of one row */
constexpr uint32_t kEdListChunk = 256u * kEdListRounds * (uint32_t)kEdNT;
__global__ __launch_bounds__(256) void k_ed_hist_list(ed_args a, const float4* __restrict__ spos,
                                                      const uint32_t* __restrict__ pending,
                                                      const uint32_t* __restrict__ n_pending, uint32_t row0,
                                                      uint32_t rows_cap, unsigned long long* __restrict__ hist) {
    __shared__ unsigned long long s_h[kEdHistWords];
    __shared__ uint32_t s_x[18];
    if (ed_stopped(a)) return;
    const uint32_t n_all = *n_pending;
    const uint32_t n_rows = min(n_all > row0 ? n_all - row0 : 0u, rows_cap);
    uint32_t* const s_cnt = reinterpret_cast<uint32_t*>(s_h);
    const uint32_t e_base = blockIdx.x * kEdListChunk;
    const uint32_t run = (n_rows + gridDim.y - 1u) / gridDim.y; /* consecutive rows per workgroup */
    /* The rows are in visit order, so a workgroup's consecutive rows are mostly hypotheses of ONE first point: the same
     * neighbourhood list. A lane's entries of that list (scene index, scan position, the point itself) stay in its
     * registers from one row to the next and are fetched again only when the neighbourhood changes: the two dependent
     * loads at the head of every row's chain (list entry -> scene position) are paid once per neighbourhood. */
    uint32_t staged_slot = 0xFFFFFFFFu;
    uint32_t sidx[kEdListRounds][kEdNT], pos[kEdListRounds][kEdNT];
    f3 pt[kEdListRounds][kEdNT];
    bool live[kEdListRounds][kEdNT];
    for (uint32_t row = blockIdx.y * run; row < min(n_rows, (blockIdx.y + 1u) * run); ++row) {
        if (!ed_row_selected(a, row0 + row)) continue;
        const uint32_t hl = pending[row0 + row];
        const uint32_t slot = a.ball.slot_of_hyp[a.h0 + hl];
        const uint32_t n_e = min(a.ball.blist_n[slot], a.ball.blist_stride);
        if (e_base >= n_e) continue;
        const pose12 T = load_pose(a.sc.poses + (size_t)(a.h0 + hl) * 16);
        const uint32_t S = a.ball.S_of_slot[slot];
        const uint32_t* tests = a.ball.tests_of_slot + 18u * slot;
        if (threadIdx.x < kEdHistWords) s_h[threadIdx.x] = threadIdx.x >= kEdHistMin ? ~0ull : 0ull;
        if (threadIdx.x >= 64u && threadIdx.x < 82u) {
            const uint32_t tk = tests[threadIdx.x - 64u];
            s_x[threadIdx.x - 64u] = tk > 0u ? tk - 1u : 0u;
        }
        __syncthreads();
        const float per_pos = S ? 20.f / (float)S : 0.f;
        if (slot != staged_slot) { /* uniform: the whole workgroup changes neighbourhood together */
            const uint2* bl = a.ball.blist + (size_t)slot * a.ball.blist_stride;
#pragma unroll
            for (uint32_t r = 0; r < kEdListRounds; ++r) {
                uint2 en[kEdNT];
#pragma unroll
                for (int i = 0; i < kEdNT; ++i) {
                    const uint32_t e = e_base + (r * (uint32_t)kEdNT + (uint32_t)i) * 256u + threadIdx.x;
                    live[r][i] = e < n_e;
                    en[i] = live[r][i] ? bl[e] : make_uint2(0u, 0u);
                }
#pragma unroll
                for (int i = 0; i < kEdNT; ++i) {
                    const float4 P = live[r][i] ? spos[en[i].x] : make_float4(0.f, 0.f, 0.f, 0.f);
                    sidx[r][i] = en[i].x;
                    pos[r][i] = en[i].y;
                    pt[r][i] = make3(P.x, P.y, P.z);
                }
            }
            staged_slot = slot;
        }
#pragma unroll
        for (uint32_t r = 0; r < kEdListRounds; ++r) {
            bool tail[kEdNT], corr[kEdNT];
            long long fx[kEdNT];
            eval_points_staged<kEdNT>(a.sc, [&](int) -> const pose12& { return T; }, pt[r], sidx[r], live[r], tail, corr, fx);
#pragma unroll
            for (int i = 0; i < kEdNT; ++i) {
                if (!tail[i]) continue;
                /* interval of pos: j = #{k : x_k <= pos}; the guess is exact up to the rounding of tests[] */
                uint32_t j = min(18u, (uint32_t)((float)(pos[r][i] + 1u) * per_pos));
                while (j < 18u && s_x[j] <= pos[r][i]) ++j;
                while (j > 0u && s_x[j - 1u] > pos[r][i]) --j;
                if (corr[i]) {
                    atomicAdd(&s_cnt[j], 1u);
                    atomicAdd(&s_h[kEdHistFx + j], (unsigned long long)fx[i]);
                    if ((unsigned long long)fx[i] >> 41) atomicAdd(&s_cnt[19], 1u);
                }
                if (j > 0u) {
                    const unsigned long long key = ((unsigned long long)pos[r][i] << 42) | (corr[i] ? 0ull : 1ull << 41) |
                                                   (corr[i] ? ((unsigned long long)fx[i] & ((1ull << 41) - 1ull)) : 0ull);
                    if (key < s_h[kEdHistMin + j - 1u]) atomicMin(&s_h[kEdHistMin + j - 1u], key);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < kEdHistWords - 1u) {
            const unsigned long long v = s_h[threadIdx.x];
            unsigned long long* g = hist + (size_t)row * kEdHistWords + threadIdx.x;
            if (threadIdx.x < kEdHistMin) {
                if (v) atomicAdd(g, v);
            } else if (v != ~0ull)
                atomicMin(g, v);
        }
        __syncthreads();
    }
}

/* the rule on the reductions of k_ed_hist_wide: one thread per pending row */
__global__ __launch_bounds__(64) void k_ed_hist_walk(ed_args a, ed_out o, float accept_thresh,
                                                     const uint32_t* __restrict__ pending,
                                                     const uint32_t* __restrict__ n_pending, uint32_t row0,
                                                     uint32_t rows_cap, const unsigned long long* __restrict__ hist,
                                                     uint32_t* __restrict__ fb_list, uint32_t* __restrict__ fb_n) {
    if (ed_stopped(a)) return;
    const uint32_t row = blockIdx.x * 64 + threadIdx.x;
    const uint32_t n_all = *n_pending;
    const uint32_t n_rows = min(n_all > row0 ? n_all - row0 : 0u, rows_cap);
    if (row >= n_rows || !ed_row_selected(a, row0 + row)) return;
    const uint32_t hl = pending[row0 + row];
    const uint32_t slot = a.ball.slot_of_hyp[a.h0 + hl];
    const uint32_t S = a.ball.S_of_slot[slot];
    const unsigned long long* h = hist + (size_t)row * kEdHistWords;
    const uint32_t* cnt = reinterpret_cast<const uint32_t*>(h);
    bool fallback = cnt[19] != 0u;
    uint32_t c = cnt[0];
    unsigned long long f = h[kEdHistFx];
    int32_t dropped = -1;
    uint32_t drop_count = 0, drop_pos = 0;
    unsigned long long drop_fx = 0;
    for (uint32_t k = 0; k < 18u && !fallback; ++k) {
        const unsigned long long key = h[kEdHistMin + k];
        if (key == ~0ull) { /* no tail position in [x_k, x_k+1): fine only if none follows either */
            for (uint32_t j = k + 1u; j < 18u; ++j) fallback = fallback || h[kEdHistMin + j] != ~0ull;
            break;
        }
        const uint32_t pos = (uint32_t)(key >> 42);
        const bool is_corr = !((key >> 41) & 1ull);
        const uint32_t corrs = c + (is_corr ? 1u : 0u);
        const uint32_t upper = early_drop_upper(pos + 1u, S, corrs);
        if ((float)upper < accept_thresh) { /* reference :500 */
            dropped = (int32_t)k;
            drop_count = corrs;
            drop_pos = pos;
            drop_fx = f + (is_corr ? (key & ((1ull << 41) - 1ull)) : 0ull);
            break;
        }
        c += cnt[k + 1u];
        f += h[kEdHistFx + k + 1u];
    }
    if (fallback) {
        fb_list[atomicAdd(fb_n, 1u)] = hl;
        return;
    }
    const uint32_t hyp = a.h0 + hl;
    if (dropped >= 0) {
        o.counts[hyp] = drop_count;
        o.dropped_at[hyp] = dropped;
        o.saved[hyp] = S - (drop_pos + 1u);
        o.limit[hyp] = drop_pos;
        a.sc.scores_fx[hyp] = drop_fx;
    } else {
        /* every interval after the last test is in c / f already, or empty */
        uint32_t ct = 0;
        unsigned long long ft = 0;
        for (uint32_t j = 0; j < 19u; ++j) {
            ct += cnt[j];
            ft += h[kEdHistFx + j];
        }
        o.counts[hyp] = ct;
        o.dropped_at[hyp] = -1;
        o.saved[hyp] = 0u;
        o.limit[hyp] = S ? S - 1u : 0xFFFFFFFFu;
        a.sc.scores_fx[hyp] = ft;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Fused variant: one workgroup does all three passes for its hypothesis with the two bit arrays in LDS
 * (LDS atomics instead of global ones, no memsets, one launch). Used whenever 2 * words * 8 bytes fit
 * next to the tile list; the three-kernel path above remains for larger subsets and as the cross-check
 * (TM_ED_UNFUSED=1).
 * ---------------------------------------------------------------------------------------------- */
constexpr int kEdFusedChunk = 2048; /* tiles per culling round in the fused kernel: 8 KB of LDS */
constexpr size_t kEdFusedLdsMax = 48 * 1024; /* dynamic LDS for the two bit arrays: subsets up to 196k points */

constexpr uint32_t kEdPrefixWordsMax = 2048; /* most scan positions a prefix-only workgroup holds bits for: 131,072
                                                (32 KiB of LDS); the host asks for what the first checkpoint needs */
/* words the prefix step needs for a scan of S positions: first checkpoint (5 %) + 1024 positions */
static inline uint32_t ed_prefix_words(uint32_t S) {
    const uint32_t need = ((uint32_t)(0.05f * (float)S) + 1024u + 63u) / 64u + 1u;
    return std::max(16u, std::min(need, kEdPrefixWordsMax));
}

/* PREFIX = true: step 0 only, with 4 KiB of static LDS for the bits (eight workgroups per CU instead of the
 * two or three the full bit arrays allow): decides the 98 % of hypotheses that drop at the first checkpoint and
 * appends the others to `pending`. PREFIX = false: the whole rule; with a pending list, workgroup b takes
 * hypothesis pending[b] (workgroups beyond *n_pending leave at once). */
template <bool PREFIX>
__global__ __launch_bounds__(256) void k_ed_fused(ed_args a, ed_cull ca, ed_out o, float accept_thresh,
                                                  const uint32_t* __restrict__ tests, const float4* __restrict__ oref,
                                                  uint32_t* __restrict__ pending, uint32_t* __restrict__ n_pending) {
    extern __shared__ unsigned long long s_dyn[]; /* [bit_words] in-grid bits, then [bit_words] correspondence bits */
    unsigned long long* s_bits = s_dyn;
    __shared__ uint32_t s_list[PREFIX ? 1 : kEdFusedChunk];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_smask;
    __shared__ uint32_t s_limit, s_count, s_saved;
    __shared__ int32_t s_dropped;
    __shared__ unsigned long long s_fx[4];
    /* the continuation of the hypotheses the prefix kernel could not decide runs on a small grid whose
     * workgroups walk the pending list; the other launches take one hypothesis per workgroup */
    if (ed_stopped(a)) return;
    const bool from_list = !PREFIX && pending != nullptr;
    const uint32_t n_items = from_list ? *n_pending : gridDim.x;
    const uint32_t* const tests_in = tests;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    uint32_t hl = from_list ? pending[item] : item;
    if (PREFIX && a.rng_starts) { /* merged step: (local index, batch) -> global hypothesis, or nothing to do */
        const uint32_t b = blockIdx.y;
        hl = a.rng_starts[b] + a.seg_step * a.seg_len + blockIdx.x;
        if (blockIdx.x >= a.seg_len || hl >= a.rng_starts[b + 1] || a.rng_stop[b] != 0xFFFFFFFFu) return;
    }
    tests = tests_in;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t bit_words = PREFIX ? a.prefix_words : a.words;
    unsigned long long* tb = s_bits;
    unsigned long long* cb = s_bits + bit_words;
    for (uint32_t w = threadIdx.x; w < 2u * bit_words; w += 256) s_bits[w] = 0ull;
    const pose12 T = load_pose(a.sc.poses + (size_t)(a.h0 + hl) * 16);
    const float rnorm = sqrtf(((T.r00 * T.r00 + T.r10 * T.r10) + (T.r20 * T.r20 + T.r01 * T.r01)) +
                              ((T.r11 * T.r11 + T.r21 * T.r21) + (T.r02 * T.r02 + T.r12 * T.r12)) +
                              T.r22 * T.r22);
    const float abs_eps = 1e-6f * (3.f * rnorm * ca.coord_max + fabsf(T.tx) + fabsf(T.ty) + fabsf(T.tz) + 1.f);
    const bool ball = a.ball.slot_of_hyp != nullptr;
    const uint32_t slot = ball ? a.ball.slot_of_hyp[a.h0 + hl] : 0u;
    const uint32_t S = ball ? a.ball.S_of_slot[slot] : a.S;
    if (ball) tests = a.ball.tests_of_slot + 18u * slot;
    f3 bc = make3(0.f, 0.f, 0.f);
    if (ball) {
        float4 C = a.ball.centre_pos[slot];
        bc = make3(C.x, C.y, C.z);
    }
    const float ball_margin = 1e-5f * a.ball.radius + 4e-6f * ca.coord_max + 1e-12f;
    const unsigned long long* member = ball ? a.ball.member + (size_t)slot * a.ball.mwords : nullptr;
    const uint32_t* wprefix = ball ? a.ball.wprefix + (size_t)slot * a.ball.mwords : nullptr;
    /* 98 % of the hypotheses of a search are dropped at the first checkpoint (5 % of the scan), so the bits are
     * produced in two steps. Step 0 takes the first B = first checkpoint + 1024 scan positions straight from
     * the neighbourhood list (position -> point, the order project_ scans in) — a few thousand points, no
     * culling of the scene at all — and the walk decides on those; only a hypothesis that is still alive
     * (or whose check needs a later position) runs step 1, the tile-culled pass over the rest of its
     * neighbourhood. The bits, hence counts, checkpoint, `saved` and score, are the same as when everything is
     * evaluated at once. */
    __shared__ uint32_t s_done;
    const uint32_t S_words = (S + 63u) / 64u;
    uint32_t B_words = (tests[0] + 1024u + 63u) / 64u;
    if (B_words > S_words) B_words = S_words;
    if (PREFIX && B_words > a.prefix_words) B_words = a.prefix_words; /* a shorter prefix may just not decide */
    if (!PREFIX && (pending || TM_ABLATE_EQ(a.sc, 30))) B_words = 0u; /* the whole scan in the tile-culled pass */
    for (uint32_t step = 0; step < (PREFIX ? 1u : 2u); ++step) {
        const uint32_t pos_lo = step == 0 ? 0u : B_words * 64u;
        const uint32_t pos_hi = step == 0 ? B_words * 64u : 0xFFFFFFFFu;
        const uint32_t walk_words = step == 0 ? B_words : a.words;
        if (step == 0) {
            if (B_words == 0u) {
                if (PREFIX) { /* nothing to decide on: hand over */
                    if (threadIdx.x == 0) pending[atomicAdd(n_pending, 1u)] = hl;
                    return;
                }
                continue;
            }
            score_args sc0 = a.sc;
            sc0.sref = oref; /* original order, as pass 3 */
            const uint32_t n_pos = min(pos_hi, S);
            if (PREFIX && ball && a.ball.plist) {
                /* the same positions in the scene's curve order (ed_ball::plist): same bits, coherent loads */
                const uint2* pl = a.ball.plist + (size_t)slot * a.ball.plist_stride;
                const uint32_t n_e = min(a.ball.plist_n[slot], a.ball.plist_stride);
                for (uint32_t e = threadIdx.x; e < n_e; e += 256) {
                    const uint2 en = pl[e];
                    if (en.y >= n_pos) continue; /* the list was cut for this slot's B; a shorter prefix takes less */
                    const float4 P = ca.spos[en.x];
                    long long fx0 = 0;
                    uint32_t pass0 = 0;
                    bool tail = false;
                    const uint32_t corr = eval_point(a.sc, T, make3(P.x, P.y, P.z), en.x, fx0, pass0, nullptr, &tail);
                    if (!tail) continue;
                    const unsigned long long bit = 1ull << (en.y & 63u);
                    atomicOr(&tb[en.y >> 6], bit);
                    if (corr) atomicOr(&cb[en.y >> 6], bit);
                }
            } else
            for (uint32_t pos = threadIdx.x; pos < n_pos; pos += 256) {
                const uint32_t idx = ball ? (uint32_t)a.ball.list[(size_t)slot * a.ball.list_stride + pos]
                                          : (a.subset ? (uint32_t)a.subset[pos] : pos);
                if (a.otan[idx].w != 0.f) continue; /* mask_ (:434) */
                const float4 P = a.opos[idx];
                long long fx0 = 0;
                uint32_t pass0 = 0;
                bool tail = false;
                const uint32_t corr = eval_point(sc0, T, make3(P.x, P.y, P.z), idx, fx0, pass0, nullptr, &tail);
                if (!tail) continue;
                const unsigned long long bit = 1ull << (pos & 63u);
                atomicOr(&tb[pos >> 6], bit);
                if (corr) atomicOr(&cb[pos >> 6], bit);
            }
            __syncthreads();
        } else if (!PREFIX)
    /* ---- pass 1: bits ---- */
    for (uint32_t t0 = 0; t0 < ca.n_tiles; t0 += kEdFusedChunk) {
        if (wave == 0) {
            const uint32_t u = (t0 >> 6) + lane;
            bool hit = false;
            if (lane < (uint32_t)(kEdFusedChunk / 64) && u * 64u < ca.n_tiles && (ball || ca.tile_bits[u])) {
                float4 sb = ca.supers[u];
                hit = sphere_hits_box(a.sc, T, rnorm, abs_eps, sb);
                if (ball) hit = hit && !(len(make3(sb.x, sb.y, sb.z) - bc) > a.ball.radius + sb.w * 1.00001f + ball_margin);
            }
            unsigned long long mask = __ballot(hit);
            if (lane == 0) {
                s_smask = mask;
                s_n = 0;
            }
        }
        __syncthreads();
        const unsigned long long smask = s_smask;
        for (uint32_t r = wave; r < (uint32_t)(kEdFusedChunk / 64); r += 4) {
            if (!((smask >> r) & 1ull)) continue;
            const uint32_t t = t0 + r * 64u + lane;
            if (t < ca.n_tiles && (ball || ((ca.tile_bits[t >> 6] >> lane) & 1ull))) {
                float4 tbs = ca.tiles[t];
                bool hit = sphere_hits_box(a.sc, T, rnorm, abs_eps, tbs);
                if (ball) hit = hit && !(len(make3(tbs.x, tbs.y, tbs.z) - bc) > a.ball.radius + tbs.w * 1.00001f + ball_margin);
                if (hit) s_list[atomicAdd(&s_n, 1u)] = t;
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        for (uint32_t i = wave; i < n; i += 4) {
            const uint32_t sidx = s_list[i] * 64u + lane;
            const float4 P = ca.spos[sidx];
            if (!(P.x == P.x)) continue; /* mask_ (:434), non-finite, padding */
            /* in-grid test first (the arithmetic of eval_point): only in-grid points have a bit, and only those
             * whose scan position belongs to this step go through the gathers */
            {
                const f3 q = xform_point(T, make3(P.x, P.y, P.z));
                const float vx = a.sc.g.sx * q.x + a.sc.g.tx, vy = a.sc.g.sy * q.y + a.sc.g.ty,
                            vz = a.sc.g.sz * q.z + a.sc.g.tz;
                const bool in = (vx > -1.f) & (vx < a.sc.g.fex) & (vy > -1.f) & (vy < a.sc.g.fey) & (vz > -1.f) &
                                (vz < a.sc.g.fez);
                if (!in) continue;
            }
            const uint32_t orig = __float_as_uint(P.w);
            uint32_t pos;
            if (ball) {
                const unsigned long long mbits = member[orig >> 6];
                if (!((mbits >> (orig & 63u)) & 1ull)) continue;
                pos = wprefix[orig >> 6] + (uint32_t)__popcll(mbits & ((1ull << (orig & 63u)) - 1ull));
            } else {
                pos = ca.pos_of_orig[orig];
                if (pos == 0xFFFFFFFFu) continue;
            }
            if (pos < pos_lo || pos >= pos_hi) continue;
            long long fx = 0;
            uint32_t pass = 0;
            bool tail = false;
            const uint32_t corr = eval_point(a.sc, T, make3(P.x, P.y, P.z), sidx, fx, pass, nullptr, &tail);
            if (!tail) continue;
            const unsigned long long bit = 1ull << (pos & 63u);
            atomicOr(&tb[pos >> 6], bit);
            if (corr) atomicOr(&cb[pos >> 6], bit);
        }
        __syncthreads();
    }
    /* ---- pass 2: the sequential rule, one wave; over the first walk_words words of the bit arrays ---- */
    if (wave == 0) {
        uint32_t k = 0, min_pos = 0;
        if (tests[0] > 0) min_pos = tests[0] - 1;
        uint32_t corr_before = 0;
        int32_t dropped = -1;
        uint32_t drop_count = 0, drop_pos = 0;
        for (uint32_t w0 = 0; w0 < walk_words && dropped < 0; w0 += 64) {
            const uint32_t w = w0 + lane;
            const unsigned long long tword = w < walk_words ? tb[w] : 0ull;
            const unsigned long long cword = w < walk_words ? cb[w] : 0ull;
            uint32_t pc = (uint32_t)__popcll(cword);
            uint32_t incl = pc;
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t v = __shfl_up(incl, off);
                if ((int)lane >= off) incl += v;
            }
            const uint32_t chunk_end = (w0 + 64) * 64;
            while (k < 18 && min_pos < chunk_end) {
                unsigned long long bits = tword;
                const uint32_t wbase = w * 64;
                if (min_pos >= wbase + 64)
                    bits = 0ull;
                else if (min_pos > wbase)
                    bits &= ~0ull << (min_pos - wbase);
                unsigned long long have = __ballot(bits != 0ull);
                if (!have) break;
                const int L = __ffsll((long long)have) - 1;
                const unsigned long long lbits = __shfl(bits, L);
                const unsigned long long lcorr = __shfl(cword, L);
                const uint32_t lincl = __shfl(incl, L);
                const uint32_t lpc = __shfl(pc, L);
                const int b = __ffsll((long long)lbits) - 1;
                const uint32_t pos = (w0 + (uint32_t)L) * 64 + (uint32_t)b;
                const unsigned long long upto = (b == 63) ? ~0ull : ((2ull << b) - 1ull);
                const uint32_t corrs = corr_before + (lincl - lpc) + (uint32_t)__popcll(lcorr & upto);
                const uint32_t upper = early_drop_upper(pos + 1, S, corrs);
                if ((float)upper < accept_thresh) { /* reference :500 */
                    dropped = (int32_t)k;
                    drop_count = corrs;
                    drop_pos = pos;
                    break;
                }
                ++k;
                min_pos = pos + 1;
                if (k < 18 && tests[k] > 0 && tests[k] - 1 > min_pos) min_pos = tests[k] - 1;
            }
            corr_before += __shfl(incl, 63);
        }
        if (lane == 0) {
            s_dropped = dropped;
            s_count = dropped >= 0 ? drop_count : corr_before;
            s_saved = dropped >= 0 ? S - (drop_pos + 1) : 0u;
            s_limit = dropped >= 0 ? drop_pos : (S ? S - 1 : 0xFFFFFFFFu);
            /* decided: dropped inside the evaluated prefix, or the prefix was the whole scan */
            s_done = (dropped >= 0 || walk_words >= S_words) ? 1u : 0u;
        }
    }
    __syncthreads();
    if (s_done) break;
    if (PREFIX) { /* not decided inside the prefix: the full kernel takes this hypothesis */
        if (threadIdx.x == 0) pending[atomicAdd(n_pending, 1u)] = hl;
        return;
    }
    } /* step */
    /* ---- pass 3: score terms of the correspondences up to the stop position ---- */
    const uint32_t lim = s_limit;
    long long fx = 0;
    if (lim != 0xFFFFFFFFu) {
        score_args sc = a.sc;
        sc.sref = oref; /* original order */
        /* one correspondence word per thread and turn: the set bits are spread evenly because the scan
         * order (ascending scene index) is unrelated to space */
        const uint32_t w_last = lim >> 6;
        for (uint32_t w = threadIdx.x; w <= w_last; w += 256) {
            unsigned long long bitsw = cb[w];
            if (w == w_last && (lim & 63u) != 63u) bitsw &= (2ull << (lim & 63u)) - 1ull;
            while (bitsw) {
                const uint32_t pos = w * 64u + (uint32_t)__builtin_ctzll(bitsw);
                bitsw &= bitsw - 1ull;
                const uint32_t idx = ball ? (uint32_t)a.ball.list[(size_t)slot * a.ball.list_stride + pos]
                                          : (a.subset ? (uint32_t)a.subset[pos] : pos);
                const float4 P = a.opos[idx];
                uint32_t pass = 0;
                (void)eval_point(sc, T, make3(P.x, P.y, P.z), idx, fx, pass, nullptr, nullptr);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) fx += __shfl_down(fx, off);
    if (lane == 0) s_fx[wave] = (unsigned long long)fx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t h = a.h0 + hl;
        a.sc.scores_fx[h] = (s_fx[0] + s_fx[1]) + (s_fx[2] + s_fx[3]);
        o.counts[h] = s_count;
        o.dropped_at[h] = s_dropped;
        o.saved[h] = s_saved;
        o.limit[h] = lim;
    }
    __syncthreads(); /* the shared state is reused by the next item */
    } /* item */
}

/* The first step of the rule for up to 16 consecutive hypotheses per workgroup. Hypotheses in visit order come in
 * runs that share their first scene point, i.e. their neighbourhood and its first-checkpoint list
 * (ed_ball::plist_p). Two things made one workgroup per hypothesis (k_ed_fused<true>) slow: every turn of its loop
 * had a few lanes whose point lands inside the voxel grid, so the whole wave went down the chain of dependent
 * gathers (brick code, voxel, model point, refs) with most lanes idle; and the bit arrays in LDS. Here
 *   phase 1  a thread takes eight list entries into registers and runs transform + in-grid test under every pose
 *            of the run (the pose is uniform over the workgroup: scalar registers; no memory access at all); the
 *            (hypothesis, entry) pairs that land in the grid are appended to a queue in LDS (wave ballots);
 *   phase 2  the queue is evaluated with all lanes busy (pose of the lane's hypothesis from LDS);
 * and checkpoint 0 only needs #{correspondences at pos < x_0} and the first evaluated position at or after x_0
 * (see k_ed_hist_wide) — two reductions per hypothesis, no bit arrays. A hypothesis the first checkpoint does not
 * drop, or whose list ends before a test position, goes to the continuation, which evaluates the whole rule. */
#ifndef TM_ED_PK
#define TM_ED_PK 4
#endif
constexpr uint32_t kEdPG = 16, kEdPK = TM_ED_PK, kEdPQ = 4096;
#ifndef TM_ED_NTQ
#define TM_ED_NTQ 2
#endif
constexpr int kEdNTQ = TM_ED_NTQ; /* queue entries a lane evaluates together */
/* BOUND: the bound pass of a search step (ed_args::bound_first) — an instantiation of its own: with it the kernel needs 142
 * registers instead of 116, a wave per SIMD less, and the scoring calls (which never take the pass) lost 14 % */
template <bool BOUND>
__global__ __launch_bounds__(256) void k_ed_prefix_group(ed_args a, ed_out o, float accept_thresh, uint32_t n_hyp,
                                                         uint32_t G, uint32_t sb_lds_words,
                                                         uint32_t* __restrict__ pending,
                                                         uint32_t* __restrict__ n_pending) {
    extern __shared__ uint32_t s_sb[]; /* score_args::super_near, when it fits */
    __shared__ uint32_t s_cnt[kEdPG], s_slot[kEdPG], s_over[kEdPG];
    __shared__ unsigned long long s_fx[kEdPG], s_min[kEdPG];
    __shared__ float s_pose[kEdPG * 16];
    __shared__ uint32_t s_q[kEdPQ];
    __shared__ uint32_t s_qn;
    __shared__ uint32_t s_near[kEdPG], s_tpos[kEdPG], s_done[kEdPG]; /* bound_first: near points before x0, test position, settled */
    if (ed_stopped(a)) return;
    uint32_t h_first, h_end;
    if (a.rng_starts) { /* merged step: batch blockIdx.y, the step's segment of it */
        const uint32_t b = blockIdx.y;
        const uint32_t lo = a.rng_starts[b] + a.seg_step * a.seg_len;
        const uint32_t hi = min(lo + a.seg_len, a.rng_starts[b + 1]);
        h_first = lo + blockIdx.x * G;
        if (blockIdx.x * G >= a.seg_len || h_first >= hi || a.rng_stop[b] != 0xFFFFFFFFu) return;
        h_end = min(h_first + G, hi);
    } else {
        h_first = blockIdx.x * G;
        if (h_first >= n_hyp) return;
        h_end = min(h_first + G, n_hyp);
    }
    const uint32_t n_g = h_end - h_first, lane = threadIdx.x & 63u;
    if (threadIdx.x < kEdPG) {
        s_cnt[threadIdx.x] = 0u;
        s_over[threadIdx.x] = 0u;
        s_fx[threadIdx.x] = 0ull;
        s_min[threadIdx.x] = ~0ull;
        s_near[threadIdx.x] = 0u;
        s_tpos[threadIdx.x] = 0xFFFFFFFFu;
        s_done[threadIdx.x] = 0u;
        s_slot[threadIdx.x] = threadIdx.x < n_g ? a.ball.slot_of_hyp[a.h0 + h_first + threadIdx.x] : 0xFFFFFFFFu;
    }
    if (threadIdx.x < n_g * 16u) s_pose[threadIdx.x] = a.sc.poses[(size_t)(a.h0 + h_first) * 16 + threadIdx.x];
    if (threadIdx.x == 0) s_qn = 0u;
    for (uint32_t w = threadIdx.x; w < sb_lds_words; w += 256u) s_sb[w] = a.sc.super_near[w];
    const uint32_t* const sbits = !a.sc.super_near ? nullptr : sb_lds_words ? s_sb : a.sc.super_near;
    __syncthreads();
    for (uint32_t g0 = 0; g0 < n_g;) {
        const uint32_t slot = s_slot[g0];
        uint32_t g1 = g0 + 1u;
        while (g1 < n_g && s_slot[g1] == slot) ++g1;
        const uint32_t S = a.ball.S_of_slot[slot];
        const uint32_t t0 = a.ball.tests_of_slot[18u * slot];
        const uint32_t x0 = t0 > 0u ? t0 - 1u : 0u;
        const uint32_t S_words = (S + 63u) / 64u;
        uint32_t B_words = (t0 + 1024u + 63u) / 64u;
        if (B_words > S_words) B_words = S_words;
        if (B_words > a.prefix_words) B_words = a.prefix_words;
        const uint32_t n_pos = min(B_words * 64u, S);
        const uint32_t n_e = min(min(a.ball.plist_n[slot], a.ball.plist_stride), 1u << 20);
        const uint2* pl = a.ball.plist + (size_t)slot * a.ball.plist_stride;
        const float4* pp = a.ball.plist_p + (size_t)slot * a.ball.plist_stride;
        /* phase 2: the queued (hypothesis, entry) pairs, all lanes busy */
        auto drain = [&]() {
            __syncthreads();
            const uint32_t n_q = s_qn;
            for (uint32_t i0 = threadIdx.x; i0 < n_q; i0 += 256u * kEdNTQ) {
                uint32_t g[kEdNTQ], sidx[kEdNTQ], pos[kEdNTQ];
                f3 pt[kEdNTQ];
                bool live[kEdNTQ], tail[kEdNTQ], corr[kEdNTQ];
                long long fx[kEdNTQ];
#pragma unroll
                for (int i = 0; i < kEdNTQ; ++i) {
                    const uint32_t qi = i0 + (uint32_t)i * 256u;
                    live[i] = qi < n_q;
                    const uint32_t ent = live[i] ? s_q[qi] : 0u, e = ent & 0xFFFFFu;
                    g[i] = ent >> 20;
                    const float4 P = live[i] ? pp[e] : make_float4(0.f, 0.f, 0.f, 0.f);
                    pos[i] = live[i] ? pl[e].y : 0u;
                    pt[i] = make3(P.x, P.y, P.z);
                    sidx[i] = __float_as_uint(P.w);
                }
                eval_points_staged<kEdNTQ>(a.sc, [&](int i) { return load_pose(s_pose + g[i] * 16u); }, pt, sidx, live, tail,
                                          corr, fx);
#pragma unroll
                for (int i = 0; i < kEdNTQ; ++i) {
                    if (!tail[i]) continue; /* live entries always are: phase 1 ran the same test */
                    if (corr[i] && ((unsigned long long)fx[i] >> 41)) s_over[g[i]] = 1u;
                    if (pos[i] < x0) {
                        if (corr[i]) {
                            atomicAdd(&s_cnt[g[i]], 1u);
                            atomicAdd(&s_fx[g[i]], (unsigned long long)fx[i]);
                        }
                    } else {
                        const unsigned long long key = ((unsigned long long)pos[i] << 42) | (corr[i] ? 0ull : 1ull << 41) |
                                                       (corr[i] ? ((unsigned long long)fx[i] & ((1ull << 41) - 1ull)) : 0ull);
                        if (key < s_min[g[i]]) atomicMin(&s_min[g[i]], key);
                    }
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) s_qn = 0u;
            __syncthreads();
        };
        if (BOUND) {
            /* The bound pass: transform + in-grid test + brick code only (the code array stays in L2; no voxel, model
             * or record line is touched). Per hypothesis: U = its NEAR points before x0 — every correspondence is a near
             * point — and the test position, the first in-grid position at or after x0 (what the exact pass finds too).
             * early_drop_upper is non-decreasing in the count, so upper(pos + 1, S, U + 1) below the threshold means
             * the exact count drops it at this checkpoint as well: settled, and with nothing of a dropped hypothesis
             * reported in a search, its exact count is never needed. The others take the exact pass below. */
            for (uint32_t e0 = 0; e0 < n_e; e0 += 256u * kEdPK) {
                float4 P[kEdPK];
                uint32_t pos[kEdPK];
#pragma unroll
                for (uint32_t k = 0; k < kEdPK; ++k) {
                    const uint32_t e = e0 + k * 256u + threadIdx.x;
                    pos[k] = 0xFFFFFFFFu;
                    P[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < n_e) {
                        P[k] = pp[e];
                        const uint32_t q = pl[e].y;
                        if (q < n_pos) pos[k] = q;
                    }
                }
                for (uint32_t g = g0; g < g1; ++g) {
                    const pose12 T = load_pose(a.sc.poses + (size_t)(a.h0 + h_first + g) * 16);
                    uint32_t near_w = 0u, tp = 0xFFFFFFFFu;
#pragma unroll
                    for (uint32_t k = 0; k < kEdPK; ++k) {
                        f3 q;
                        uint32_t brick, lin;
                        const bool in = ev_grid(a.sc, T, make3(P[k].x, P[k].y, P[k].z), q, brick, lin) && pos[k] != 0xFFFFFFFFu;
                        bool near = in && !(sbits && !((sbits[brick >> 11] >> ((brick >> 6) & 31u)) & 1u));
                        if (near && a.sc.bcode) near = (uint32_t)a.sc.bcode[brick] <= a.sc.code_max;
                        near_w += (uint32_t)__popcll(__ballot(near && pos[k] < x0));
                        if (in && pos[k] >= x0) tp = min(tp, pos[k]);
                    }
                    for (int off = 32; off > 0; off >>= 1) tp = min(tp, (uint32_t)__shfl_xor((int)tp, off));
                    if (lane == 0u) {
                        if (near_w) atomicAdd(&s_near[g], near_w);
                        if (tp != 0xFFFFFFFFu) atomicMin(&s_tpos[g], tp);
                    }
                }
            }
            __syncthreads();
            if (threadIdx.x >= g0 && threadIdx.x < g1 && s_tpos[threadIdx.x] != 0xFFFFFFFFu) {
                const uint32_t g = threadIdx.x, tp = s_tpos[g];
                if ((float)early_drop_upper(tp + 1u, S, s_near[g] + 1u) < accept_thresh) {
                    const uint32_t hyp = a.h0 + h_first + g;
                    o.counts[hyp] = 0u; /* dropped at the first checkpoint; its count is not known and not asked for */
                    o.dropped_at[hyp] = 0;
                    o.saved[hyp] = S - (tp + 1u);
                    o.limit[hyp] = tp;
                    a.sc.scores_fx[hyp] = 0ull;
                    s_done[g] = 1u;
                }
            }
            __syncthreads();
            bool all_done = true;
            for (uint32_t g = g0; g < g1; ++g) all_done = all_done && s_done[g] != 0u;
            if (all_done) {
                g0 = g1;
                continue;
            }
        }
        for (uint32_t e0 = 0; e0 < n_e; e0 += 256u * kEdPK) {
            float4 P[kEdPK];
            uint32_t pos[kEdPK]; /* 0xFFFFFFFF = no entry */
#pragma unroll
            for (uint32_t k = 0; k < kEdPK; ++k) {
                const uint32_t e = e0 + k * 256u + threadIdx.x;
                pos[k] = 0xFFFFFFFFu;
                P[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < n_e) {
                    P[k] = pp[e];
                    const uint32_t q = pl[e].y;
                    if (q < n_pos) pos[k] = q; /* the list was cut for this slot's B; a shorter prefix takes less */
                }
            }
            for (uint32_t g = g0; g < g1; ++g) {
                if (BOUND && s_done[g]) continue; /* settled by the bound pass (uniform over the workgroup) */
                const pose12 T = load_pose(a.sc.poses + (size_t)(a.h0 + h_first + g) * 16);
#pragma unroll
                for (uint32_t k = 0; k < kEdPK; ++k) {
                    /* eval_point up to its in-grid test; a point in a super-brick without any brick in reach of the
                     * surface is evaluated ("tail") but never a correspondence: it can only be the test position */
                    f3 q;
                    uint32_t brick, lin;
                    bool in = ev_grid(a.sc, T, make3(P[k].x, P[k].y, P[k].z), q, brick, lin) && pos[k] != 0xFFFFFFFFu;
                    if (in && sbits && !((sbits[brick >> 11] >> ((brick >> 6) & 31u)) & 1u)) {
                        in = false;
                        if (pos[k] >= x0) {
                            const unsigned long long key = ((unsigned long long)pos[k] << 42) | (1ull << 41);
                            if (key < s_min[g]) atomicMin(&s_min[g], key);
                        }
                    }
                    const unsigned long long m = __ballot(in);
                    if (m) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&s_qn, (uint32_t)__popcll(m));
                        base = __shfl(base, 0);
                        if (in) s_q[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (g << 20) | (e0 + k * 256u + threadIdx.x);
                    }
                }
                /* one pose adds at most 2048 entries */
                __syncthreads();
                const uint32_t queued = s_qn;
                __syncthreads(); /* everybody has read it before the next pose appends */
                if (queued > kEdPQ - 256u * kEdPK) drain();
            }
        }
        drain();
        g0 = g1;
    }
    if (threadIdx.x < n_g && !(BOUND && s_done[threadIdx.x])) {
        const uint32_t g = threadIdx.x, hl = h_first + g, hyp = a.h0 + hl;
        const unsigned long long key = s_min[g];
        bool decided = false;
        if (key != ~0ull && !s_over[g]) {
            const uint32_t S = a.ball.S_of_slot[s_slot[g]];
            const uint32_t pos = (uint32_t)(key >> 42);
            const bool is_corr = !((key >> 41) & 1ull);
            const uint32_t corrs = s_cnt[g] + (is_corr ? 1u : 0u);
            const uint32_t upper = early_drop_upper(pos + 1u, S, corrs);
            if ((float)upper < accept_thresh) { /* reference :500: dropped at the first checkpoint */
                o.counts[hyp] = corrs;
                o.dropped_at[hyp] = 0;
                o.saved[hyp] = S - (pos + 1u);
                o.limit[hyp] = pos;
                a.sc.scores_fx[hyp] = s_fx[g] + (is_corr ? (key & ((1ull << 41) - 1ull)) : 0ull);
                decided = true;
            }
        }
        if (!decided) pending[atomicAdd(n_pending, 1u)] = hl;
    }
}

/* The pending list of one step in visit order: ascending hypothesis index (the prefix kernel appends in no order),
 * and per row its position among the pending rows of its batch | batch << 20 (ed_args::row_info). One workgroup.
 * Up to kPendingSortMax rows: rank sort in LDS. A longer list (many batches per step, a large search_segment, scenes
 * on which most hypotheses pass the first checkpoint): the values are distinct hypothesis indices below `vmax`, so
 * one bit per index in `scratch` (2 * ceil(vmax / 32) words: bits, then their exclusive prefix counts) sorts them —
 * same list, same row_info. (The first version left a long list unsorted with row_info = 0, which also zeroed the
 * batch id: the rows of every batch then followed batch 0's stop word.) */
constexpr uint32_t kPendingSortMax = 4096;
__device__ __forceinline__ uint32_t pending_batch_of(const uint32_t* __restrict__ rng_starts, uint32_t n_ranges,
                                                     uint32_t v) {
    /* last batch whose first hypothesis is <= v (empty batches share a start: the last of them is never the owner
     * unless it is the one that holds v, because an empty batch's start equals the next batch's) */
    uint32_t lo = 0, hi = n_ranges; /* invariant: rng_starts[lo] <= v, hi = first candidate known to start beyond v */
    while (lo + 1 < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (rng_starts[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(1024) void k_pending_sort(uint32_t* __restrict__ pending,
                                                       const uint32_t* __restrict__ n_pending,
                                                       const uint32_t* __restrict__ rng_starts, uint32_t n_ranges,
                                                       uint32_t cap, uint32_t* __restrict__ row_info, uint32_t vmax,
                                                       uint32_t* __restrict__ scratch) {
    __shared__ uint32_t s_h[kPendingSortMax];
    __shared__ uint32_t s_run;
    const uint32_t n = min(*n_pending, cap);
    if (n > kPendingSortMax) {
        const uint32_t W = (vmax + 31u) / 32u;
        uint32_t* bits = scratch;
        uint32_t* pre = scratch + W;
        for (uint32_t w = threadIdx.x; w < W; w += 1024) bits[w] = 0u;
        if (threadIdx.x == 0) s_run = 0u;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 1024) {
            const uint32_t v = pending[i];
            if (v < vmax) atomicOr(&bits[v >> 5], 1u << (v & 31u));
        }
        __syncthreads();
        for (uint32_t base = 0; base < W; base += 1024) { /* exclusive prefix counts of the words, 1024 at a time */
            const uint32_t w = base + threadIdx.x;
            const uint32_t cnt = w < W ? (uint32_t)__popc(bits[w]) : 0u;
            s_h[threadIdx.x] = cnt;
            __syncthreads();
            for (uint32_t off = 1; off < 1024; off <<= 1) {
                const uint32_t t = threadIdx.x >= off ? s_h[threadIdx.x - off] : 0u;
                __syncthreads();
                s_h[threadIdx.x] += t;
                __syncthreads();
            }
            const uint32_t run = s_run;
            if (w < W) pre[w] = run + s_h[threadIdx.x] - cnt;
            __syncthreads();
            if (threadIdx.x == 1023) s_run = run + s_h[1023];
            __syncthreads();
        }
        for (uint32_t w = threadIdx.x; w < W; w += 1024) {
            uint32_t word = bits[w], r = pre[w];
            while (word) {
                const uint32_t bit = (uint32_t)__ffs((int)word) - 1u;
                word &= word - 1u;
                const uint32_t v = w * 32u + bit;
                uint32_t b = 0, before_batch = 0;
                if (rng_starts) {
                    b = pending_batch_of(rng_starts, n_ranges, v);
                    const uint32_t lo = rng_starts[b];
                    before_batch = pre[lo >> 5] + (uint32_t)__popc(bits[lo >> 5] & ((1u << (lo & 31u)) - 1u));
                }
                pending[r] = v;
                row_info[r] = min(r - before_batch, 0xFFFFFu) | (b << 20);
                ++r;
            }
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < n; i += 1024) s_h[i] = pending[i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const uint32_t v = s_h[i];
        uint32_t b = 0, lo = 0;
        if (rng_starts) {
            b = pending_batch_of(rng_starts, n_ranges, v);
            lo = rng_starts[b];
        }
        uint32_t rank = 0, before_batch = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t w = s_h[j];
            rank += w < v ? 1u : 0u;
            before_batch += w < lo ? 1u : 0u;
        }
        pending[rank] = v;
        row_info[rank] = min(rank - before_batch, 0xFFFFFu) | (b << 20);
    }
}

/* what decides, after a wave of the continuation, whether a batch has its sufficient hypothesis */
struct ed_seg_check {
    const uint32_t* counts = nullptr;
    const unsigned long long* fx = nullptr;
    float sufficient = 0.f;
    uint32_t* found = nullptr;      /* one word per batch (merged step) or the batch's word */
    const uint32_t* starts = nullptr; /* merged step: device hypothesis offsets of the batches */
    uint32_t n_ranges = 0, seg_len = 0, step = 0;
    uint32_t h0 = 0, h1 = 0;        /* single batch: the segment's hypotheses */
};

/* first hypothesis of [n0, n1) with a positive score and (float)count > sufficient -> *found, unless an earlier
 * segment already has one; one workgroup */
__global__ __launch_bounds__(1024) void k_segment_first(const uint32_t* __restrict__ counts,
                                                        const unsigned long long* __restrict__ fx, uint32_t n0,
                                                        uint32_t n1, float sufficient, uint32_t* found) {
    __shared__ uint32_t s_first;
    if (*found != 0xFFFFFFFFu) return;
    if (threadIdx.x == 0) s_first = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t h = n0 + threadIdx.x; h < n1; h += blockDim.x)
        if (fx[h] != 0ull && (float)counts[h] > sufficient) {
            atomicMin(&s_first, h);
            break;
        }
    __syncthreads();
    if (threadIdx.x == 0 && s_first != 0xFFFFFFFFu) *found = s_first;
}
/* the same for one segment step of several batches: workgroup b looks at hypotheses
 * [starts[b] + step * seg_len, min(+ seg_len, starts[b + 1])) unless found[b] is already set */
__global__ __launch_bounds__(1024) void k_segment_first_ranges(const uint32_t* __restrict__ counts,
                                                               const unsigned long long* __restrict__ fx,
                                                               const uint32_t* __restrict__ starts, uint32_t seg_len,
                                                               uint32_t step, float sufficient, uint32_t* found) {
    __shared__ uint32_t s_first;
    const uint32_t b = blockIdx.x;
    if (found[b] != 0xFFFFFFFFu) return;
    const uint32_t n0 = starts[b] + step * seg_len, n1 = min(n0 + seg_len, starts[b + 1]);
    if (threadIdx.x == 0) s_first = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t h = n0 + threadIdx.x; h < n1; h += blockDim.x)
        if (fx[h] != 0ull && (float)counts[h] > sufficient) {
            atomicMin(&s_first, h);
            break;
        }
    __syncthreads();
    if (threadIdx.x == 0 && s_first != 0xFFFFFFFFu) found[b] = s_first;
}

/* the fused rule in two launches: the prefix kernel (small LDS, high occupancy) decides the hypotheses that
 * drop at the first checkpoint, the full kernel continues with the rest (TM_ED_ONE_KERNEL=1: full kernel only) */
static hipError_t launch_ed_fused(tm_scene* s, tm_ctx* c, uint32_t n_hyp, size_t fused_lds, const ed_args& a,
                                  const ed_cull& cu, const ed_out& o, float accept_thresh, const uint32_t* d_tests,
                                  uint32_t rows_cap = 0, uint32_t n_ranges = 0, uint32_t* d_n_step = nullptr,
                                  const ed_seg_check* check = nullptr) {
    /* n_ranges > 0: one segment step of n_ranges batches (a.rng_*): n_hyp = n_ranges * a.seg_len workgroups, the
     * pending counter d_n_step was zeroed by the caller for all steps at once */
    auto run_check = [&]() {
        if (check && check->starts)
            hipLaunchKernelGGL(k_segment_first_ranges, dim3(check->n_ranges), dim3(1024), 0, c->stream, check->counts,
                               check->fx, check->starts, check->seg_len, check->step, check->sufficient, check->found);
        else if (check)
            hipLaunchKernelGGL(k_segment_first, dim3(1), dim3(1024), 0, c->stream, check->counts, check->fx, check->h0,
                               check->h1, check->sufficient, check->found);
    };
    if (c->opt.ed_one_kernel) {
        hipLaunchKernelGGL(k_ed_fused<false>, dim3(n_hyp), dim3(256), fused_lds, c->stream, a, cu, o, accept_thresh,
                           d_tests, s->oref.p, (uint32_t*)nullptr, (uint32_t*)nullptr);
        run_check();
        return hipGetLastError();
    }
    /* the pending list holds hypothesis indices: local ones (< n_hyp) or, in a merged step, global ones (< a.sc.n_hyp) */
    const uint32_t sort_vmax = std::max(n_hyp, a.sc.n_hyp);
    uint32_t* d_n = d_n_step;
    if (!d_n) {
        if (s->ed_pending.ensure((size_t)n_hyp + 1) != TM_OK) return hipErrorOutOfMemory;
        d_n = s->ed_pending.p + n_hyp;
        hipError_t e = hipMemsetAsync(d_n, 0, 4, c->stream);
        if (e != hipSuccess) return e;
    }
    if (c->opt.ed_prefix_group && a.ball.slot_of_hyp && a.ball.plist_p && a.S < (1u << kEdHistPosBits)) {
        /* hypotheses per workgroup: as many as keep ~16k workgroups in the launch (65,536 hypotheses: 7.2 ms per batch
         * with 16, 6.7 ms with 4; a search step has 14k hypotheses) */
        const uint32_t G = c->opt.ed_prefix_group > 1 ? std::min<uint32_t>((uint32_t)c->opt.ed_prefix_group, kEdPG)
                                                      : std::max(1u, std::min(kEdPG, n_hyp / 16384u));
        const dim3 ggrid = n_ranges ? dim3((a.seg_len + G - 1u) / G, n_ranges) : dim3((n_hyp + G - 1u) / G);
        const uint32_t sb_words = a.sc.super_near ? (uint32_t)(((size_t)(a.sc.n_super + 255u) / 256u) * 8u) : 0u;
        const uint32_t sb_lds = sb_words * 4u <= 32u * 1024u ? sb_words : 0u;
        ed_args ab = a;
        /* a search step whose `sufficient` is not below the rule's own threshold: a dropped hypothesis has
         * count <= upper(count) < accept_thresh <= sufficient, so it can never be the first sufficient one and nothing
         * else of it is reported (ed_args::bound_first). A caller's smaller `sufficient` takes the exact pass. */
        ab.bound_first = (check != nullptr && c->opt.ed_bound_first && check->sufficient >= accept_thresh) ? 1u : 0u;
        if (ab.bound_first)
            hipLaunchKernelGGL(k_ed_prefix_group<true>, ggrid, dim3(256), (size_t)sb_lds * 4u, c->stream, ab, o, accept_thresh,
                               n_ranges ? 0u : n_hyp, G, sb_lds, s->ed_pending.p, d_n);
        else
            hipLaunchKernelGGL(k_ed_prefix_group<false>, ggrid, dim3(256), (size_t)sb_lds * 4u, c->stream, ab, o, accept_thresh,
                               n_ranges ? 0u : n_hyp, G, sb_lds, s->ed_pending.p, d_n);
    } else {
        const dim3 pgrid = n_ranges ? dim3(a.seg_len, n_ranges) : dim3(n_hyp);
        hipLaunchKernelGGL(k_ed_fused<true>, pgrid, dim3(256), 2 * (size_t)a.prefix_words * 8, c->stream, a, cu, o,
                           accept_thresh, d_tests, s->oref.p, s->ed_pending.p, d_n);
    }
    const bool hist = c->opt.ed_hist && a.ball.slot_of_hyp && !c->opt.ed_narrow && fused_lds <= kEdFusedLdsMax &&
                      a.S < (1u << kEdHistPosBits);
    if (hist) {
        /* continuation on interval reductions (k_ed_hist_wide): no bit rows, so every pending row fits one pass */
        ed_args aw = a;
        const uint32_t gs = (cu.n_tiles + 63u) / 64u;
        const uint32_t wave_rows = check ? (uint32_t)c->opt.ed_wave_rows : 0u;
        /* rows in visit order: the waves below need it, and consecutive rows are hypotheses of one first point with
         * near-identical poses — a workgroup of k_ed_hist_list takes a run of them and finds the lines of the
         * previous one in L1 / L2 */
        if (s->ed_row_info.ensure(n_hyp) != TM_OK || s->ed_sort_scratch.ensure(2 * (size_t)((sort_vmax + 31u) / 32u)) != TM_OK)
            return hipErrorOutOfMemory;
        hipLaunchKernelGGL(k_pending_sort, dim3(1), dim3(1024), 0, c->stream, s->ed_pending.p, d_n, a.rng_starts, n_ranges,
                           n_hyp, s->ed_row_info.p, sort_vmax, s->ed_sort_scratch.p);
        if (wave_rows) aw.row_info = s->ed_row_info.p;
        if (s->ed_hist.ensure((size_t)n_hyp * kEdHistWords) != TM_OK || s->ed_fb.ensure((size_t)n_hyp + 2) != TM_OK)
            return hipErrorOutOfMemory;
        uint32_t* fb_n = s->ed_fb.p + n_hyp;
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
        /* rows side by side: a search step has a few dozen pending rows per wave, the whole batch over a thousand */
        /* (fewer, longer workgroups — 128 / 64 / 32 — reuse more of a neighbourhood's registers and are 2 / 5 / 12 %
         * slower on the whole batch; 512 and 1024 measure the same as 256) */
        const uint32_t gy = std::min(n_hyp, check ? 64u : 256u);
        hipLaunchKernelGGL(k_ed_hist_init, dim3((uint32_t)(((size_t)n_hyp * kEdHistWords + 255) / 256)), dim3(256), 0,
                           c->stream, s->ed_hist.p, d_n, 0u, n_hyp, fb_n);
        for (uint32_t wave = 0; wave < (wave_rows ? 2u : 1u); ++wave) {
            if (wave_rows) {
                aw.ord_lo = wave == 0 ? 0u : wave_rows;
                aw.ord_hi = wave == 0 ? wave_rows : 0xFFFFFFFFu;
            }
            if (aw.ball.blist)
                hipLaunchKernelGGL(k_ed_hist_list, dim3(std::max(1u, (a.S + kEdListChunk - 1u) / kEdListChunk), gy), dim3(256), 0, c->stream,
                                   aw, cu.spos, s->ed_pending.p, d_n, 0u, n_hyp, s->ed_hist.p);
            else
                hipLaunchKernelGGL(k_ed_hist_wide, dim3(gs, gy), dim3(256), 0, c->stream, aw, cu, s->ed_pending.p, d_n, 0u,
                                   n_hyp, s->ed_hist.p);
            hipLaunchKernelGGL(k_ed_hist_walk, dim3((n_hyp + 63u) / 64u), dim3(64), 0, c->stream, aw, o, accept_thresh,
                               s->ed_pending.p, d_n, 0u, n_hyp, s->ed_hist.p, s->ed_fb.p, fb_n + wave);
            /* rows the reductions cannot decide (an empty interval before a later test position): bit arrays in LDS */
            ed_args af = a;
            af.row_info = nullptr;
            hipLaunchKernelGGL(k_ed_fused<false>, dim3(std::min<uint32_t>(n_hyp, (uint32_t)cus)), dim3(256), fused_lds,
                               c->stream, af, cu, o, accept_thresh, d_tests, s->oref.p, s->ed_fb.p, fb_n + wave);
            run_check();
        }
        if (c->opt.debug_ed) {
            uint32_t h_fb[2] = {0u, 0u}, n_pend = 0;
            (void)hipMemcpyAsync(h_fb, fb_n, 8, hipMemcpyDeviceToHost, c->stream);
            (void)hipMemcpyAsync(&n_pend, d_n, 4, hipMemcpyDeviceToHost, c->stream);
            (void)hipStreamSynchronize(c->stream);
            fprintf(stderr, "[early drop] interval reductions: %u pending, %u rows to the bit-array kernel\n", n_pend,
                    h_fb[0] + h_fb[1]);
        }
        return hipGetLastError();
    }
    if (rows_cap && a.ball.slot_of_hyp && a.tail_bits && a.corr_bits && !c->opt.ed_narrow) {
        /* wide continuation (ball mode: the scan order is the neighbourhood list): bits of all positions of the
         * pending hypotheses, then the rule and the score terms on those rows; rows_cap rows of bit arrays */
        ed_args aw = a;
        const uint32_t gw = (a.words + 255u) / 256u, gs = (cu.n_tiles + 63u) / 64u, gb = (a.words * 8u + 255u) / 256u;
        /* Only the first sufficient hypothesis of a batch counts (check != null: a search step). The rows go through
         * in two waves in visit order — the first `ed_wave_rows` pending hypotheses of every batch, then the rest of
         * the batches that still have none: a segment that contains a true match holds dozens of pending hypotheses
         * which each scan their whole neighbourhood, and the first few settle it. Whole-batch scoring (check == null)
         * reports every hypothesis: one wave. */
        const uint32_t wave_rows = check && rows_cap >= n_hyp ? (uint32_t)c->opt.ed_wave_rows : 0u;
        if (wave_rows) {
            if (s->ed_row_info.ensure(n_hyp) != TM_OK ||
                s->ed_sort_scratch.ensure(2 * (size_t)((sort_vmax + 31u) / 32u)) != TM_OK)
                return hipErrorOutOfMemory;
            hipLaunchKernelGGL(k_pending_sort, dim3(1), dim3(1024), 0, c->stream, s->ed_pending.p, d_n, a.rng_starts, n_ranges,
                               n_hyp, s->ed_row_info.p, sort_vmax, s->ed_sort_scratch.p);
            aw.row_info = s->ed_row_info.p;
        }
        ed_args as = aw; /* k_ed_score addresses scene points by original index */
        as.sc.sref = s->oref.p;
        for (uint32_t row0 = 0; row0 < n_hyp; row0 += rows_cap) {
            const uint32_t rows = std::min(rows_cap, n_hyp - row0);
            const uint32_t gy = std::min(rows, 128u);
            hipLaunchKernelGGL(k_ed_zero_rows, dim3(gw, gy), dim3(256), 0, c->stream, a.tail_bits, a.corr_bits, a.words,
                               d_n, row0, rows);
            for (uint32_t wave = 0; wave < (wave_rows ? 2u : 1u); ++wave) {
                if (wave_rows) {
                    aw.ord_lo = as.ord_lo = wave == 0 ? 0u : wave_rows;
                    aw.ord_hi = as.ord_hi = wave == 0 ? wave_rows : 0xFFFFFFFFu;
                }
                hipLaunchKernelGGL(k_ed_bits_wide, dim3(gs, gy), dim3(256), 0, c->stream, aw, cu, s->ed_pending.p, d_n, row0,
                                   rows);
                hipLaunchKernelGGL(k_ed_walk, dim3(rows), dim3(64), 0, c->stream, aw, o, accept_thresh, d_tests,
                                   s->ed_pending.p, d_n, row0);
                hipLaunchKernelGGL(k_ed_score_rows, dim3(gb, gy), dim3(256), 0, c->stream, as, o.limit, s->ed_pending.p, d_n,
                                   row0, rows);
                run_check();
            }
        }
        return hipGetLastError();
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
    hipLaunchKernelGGL(k_ed_fused<false>, dim3(std::min<uint32_t>(n_hyp, (uint32_t)cus * 2u)), dim3(256), fused_lds,
                       c->stream, a, cu, o, accept_thresh, d_tests, s->oref.p, s->ed_pending.p, d_n);
    run_check();
    return hipGetLastError();
}

/* score_args::super_near: bit sb = some brick of super-brick sb has code <= code_max; a wave = 64 super-bricks */
__global__ __launch_bounds__(256) void k_super_near(const unsigned char* __restrict__ bcode, uint32_t n_super,
                                                    uint32_t code_max, uint32_t* __restrict__ bits) {
    const uint32_t sb = blockIdx.x * 256 + threadIdx.x;
    bool near = false;
    if (sb < n_super) {
        const uint4* p = reinterpret_cast<const uint4*>(bcode + (size_t)sb * 64);
        for (int i = 0; i < 4; ++i) {
            const uint4 v = p[i];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < 4; ++j)
                for (int b = 0; b < 4; ++b) near = near || ((w[j] >> (8 * b)) & 0xFFu) <= code_max;
        }
    }
    const unsigned long long m = __ballot(near);
    if ((threadIdx.x & 63u) == 0u) {
        bits[sb >> 5] = (uint32_t)m;
        bits[(sb >> 5) + 1u] = (uint32_t)(m >> 32);
    }
}

}  // namespace tmx

using namespace tmx;

/* the super-brick bits that go with sc.bcode / sc.code_max (built once per threshold) */
static int ed_super_filter(tm_model* m, score_args& sc) {
    sc.super_near = nullptr;
    if (!sc.bcode || sc.bcode != m->brick_code.p) return TM_OK;
    const uint32_t n_super = (uint32_t)(m->voxel_slots / 4096u);
    const size_t words = ((size_t)(n_super + 255u) / 256u) * 8u;
    if (m->super_near_code != sc.code_max || m->super_near_src != sc.bcode) {
        TM_FORWARD(m->super_near.ensure(words));
        hipLaunchKernelGGL(k_super_near, dim3((n_super + 255u) / 256u), dim3(256), 0, m->ctx->stream, sc.bcode, n_super,
                           sc.code_max, m->super_near.p);
        TM_HIP_CHECK(hipGetLastError());
        m->super_near_code = sc.code_max;
        m->super_near_src = sc.bcode;
    }
    sc.super_near = m->super_near.p;
    sc.n_super = n_super;
    return TM_OK;
}

/* brick near-filter for the correspondence part of eval_point (same bound as the scoring kernel): a voxel
 * whose brick code exceeds code_max has no model point within thres; Euclidean metrics only */
static void ed_brick_filter(const tm_model* m, float thres, score_args& sc) {
    sc.bcode = nullptr;
    sc.bcode_tile = nullptr;
    sc.code_max = 255u;
    if (m->metric_kind > TM_TRAITS_PLANE || !(thres >= 0.f)) return;
    const float cell_max = std::max(std::max(1.f / m->grid.sx, 1.f / m->grid.sy), 1.f / m->grid.sz);
    const float slack = 1.7320508f * cell_max * 1.0001f + thres * 1e-5f + 1e-7f;
    const float cm = std::floor((thres + slack) / m->brick_q) + 1.f;
    if (cm < 255.f) {
        sc.bcode = m->brick_code.p;
        sc.bcode_tile = m->brick_code.p;
        sc.code_max = (uint32_t)cm;
    }
}

extern "C" int tm_score_hypotheses_early_drop(tm_scene* s, tm_model* m, const float* poses16,
                                              uint32_t n_hyp, const int32_t* subset, uint32_t n_subset,
                                              float dist_thres, float accept_prob, uint32_t* counts,
                                              double* scores, int32_t* dropped_at, uint32_t* saved) {
    TM_GUARD(s);
    TM_REQUIRE(s && m, TM_ERR_INVALID, "null scene or model");
    TM_REQUIRE(s->ctx == m->ctx, TM_ERR_INVALID, "scene and model belong to different contexts");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    TM_REQUIRE(counts && scores && dropped_at && saved, TM_ERR_INVALID, "null outputs");
    if (n_hyp == 0) return TM_OK;
    TM_REQUIRE(poses16, TM_ERR_INVALID, "null poses");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t S = subset ? n_subset : s->n;
    if (subset)
        for (uint32_t k = 0; k < n_subset; ++k)
            TM_REQUIRE(subset[k] >= 0 && (uint32_t)subset[k] < s->n, TM_ERR_INVALID,
                       "subset index out of range");
    if (S == 0) {
        for (uint32_t h = 0; h < n_hyp; ++h) {
            counts[h] = 0;
            scores[h] = 0.0;
            dropped_at[h] = -1;
            saved[h] = 0;
        }
        return TM_OK;
    }
    /* checkpoints: tests[i] = (uint32)(0.05f * (i + 1) * |subset|), reference :422-426 */
    uint32_t tests[18];
    {
        constexpr float step_size = 0.05f;
        for (uint32_t i = 0; i < 18; ++i) tests[i] = (uint32_t)(step_size * (i + 1) * S);
    }
    const uint32_t words = (S + 63) / 64;
    const uint64_t budget = 256ull << 20; /* bytes per bit array */
    uint32_t chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_hyp, budget / ((uint64_t)words * 8)));
    chunk = std::min(chunk, 65535u);
    /* buffers live on the scene and are reused across calls */
    dbuf<float>& d_poses = s->w_poses;
    dbuf<int32_t>&d_subset = s->ed_subset, &d_dropped = s->ed_dropped;
    dbuf<uint32_t>&d_tests = s->ed_tests, &d_counts = s->w_counts, &d_saved = s->ed_saved, &d_limit = s->ed_limit,
                  &d_pos = s->ed_pos;
    dbuf<unsigned long long>&d_tail = s->ed_tail, &d_corr = s->ed_corr, &d_fx = s->w_scores_fx;
    int rc = TM_OK;
    auto cleanup = [&]() {};
    if ((rc = d_poses.ensure((size_t)n_hyp * 16)) || (rc = d_tests.ensure(20)) || (rc = d_counts.ensure(n_hyp)) ||
        (rc = d_saved.ensure(n_hyp)) || (rc = d_limit.ensure(n_hyp)) || (rc = d_dropped.ensure(n_hyp)) ||
        (rc = d_fx.ensure(n_hyp)) || (rc = d_tail.ensure((size_t)chunk * words)) ||
        (rc = d_corr.ensure((size_t)chunk * words)) || (rc = d_pos.ensure(s->n)) || (rc = s->ed_tilebits.ensure((s->n_tiles + 63) / 64)) ||
        (subset && (rc = d_subset.ensure(S))))
        return rc;
#define ED_HIP(expr)                                                      \
    do {                                                                  \
        hipError_t _e = (expr);                                           \
        if (_e != hipSuccess) {                                           \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));     \
            cleanup();                                                    \
            return TM_ERR_HIP;                                            \
        }                                                                 \
    } while (0)
    ED_HIP(hipMemcpyAsync(d_poses.p, poses16, (size_t)n_hyp * 64, hipMemcpyHostToDevice, c->stream));
    ED_HIP(hipMemcpyAsync(d_tests.p, tests, sizeof(tests), hipMemcpyHostToDevice, c->stream));
    if (subset) ED_HIP(hipMemcpyAsync(d_subset.p, subset, (size_t)S * 4, hipMemcpyHostToDevice, c->stream));
    ED_HIP(hipMemsetAsync(d_fx.p, 0, (size_t)n_hyp * 8, c->stream));
    ed_args a;
    memset(&a.ball, 0, sizeof a.ball);
    a.sc.spos = nullptr;
    a.sc.sref = s->oref.p;
    a.sc.mpos = m->pos.p;
    a.sc.mpos3 = m->pos3.p;
    a.sc.mref = m->ref.p;
    a.sc.vox = m->voxels.p;
    a.sc.active = nullptr;
    a.sc.ablate = c->opt.ablate == 30 ? 30 : 0; /* TM_ABLATE=30: one-step evaluation of the fused kernel (timing comparison) */
    a.sc.kind = m->metric_kind;
    a.sc.radius = m->metric_radius;
    a.sc.dmax = dist_thres * m->resolution;
    a.sc.bcode = nullptr; /* bx/by still address the bricked voxel table */
    a.sc.bcode_tile = nullptr;
    a.sc.bx = m->bdim[0];
    a.sc.by = m->bdim[1];
    a.sc.code_max = 255u;
    a.sc.poses = d_poses.p;
    a.sc.counts = d_counts.p;
    a.sc.scores_fx = d_fx.p;
    a.sc.stat_slots = nullptr;
    a.sc.g = m->grid;
    a.sc.d2max = sq_threshold(dist_thres * m->resolution);
    a.sc.n_hyp = n_hyp;
    a.sc.hyp_per_block = 1;
    a.opos = s->opos.p;
    a.otan = s->otan.p;
    a.subset = subset ? d_subset.p : nullptr;
    a.S = S;
    a.words = words;
    a.tail_bits = d_tail.p;
    a.corr_bits = d_corr.p;
    a.stop = nullptr;
    a.rng_starts = nullptr;
    a.rng_stop = nullptr;
    a.seg_len = a.seg_step = 0;
    a.row_info = nullptr;
    a.ord_lo = 0;
    a.ord_hi = 0xFFFFFFFFu;
    a.prefix_words = ed_prefix_words(a.S);
    ed_out o;
    o.counts = d_counts.p;
    o.dropped_at = d_dropped.p;
    o.saved = d_saved.p;
    o.limit = d_limit.p;
    const float accept_thresh = accept_prob * (float)m->n; /* accept_prob * size() in float */
    /* scan position of every scene point; a subset that names a point twice (or TM_ED_BRUTEFORCE=1)
     * takes the one-thread-per-position kernel */
    bool culled = !c->opt.ed_bruteforce;
    if (culled) {
        uint32_t dup = 0;
        ED_HIP(hipMemsetAsync(d_pos.p, 0xFF, (size_t)s->n * 4, c->stream));
        ED_HIP(hipMemsetAsync(d_tests.p + 18, 0, 4, c->stream));
        hipLaunchKernelGGL(k_ed_positions, dim3((S + 255) / 256), dim3(256), 0, c->stream, a.subset, S, d_pos.p);
        if (subset) {
            hipLaunchKernelGGL(k_ed_check_positions, dim3((S + 255) / 256), dim3(256), 0, c->stream, a.subset, S,
                               d_pos.p, d_tests.p + 18);
            ED_HIP(hipMemcpyAsync(&dup, d_tests.p + 18, 4, hipMemcpyDeviceToHost, c->stream));
            ED_HIP(hipStreamSynchronize(c->stream));
        }
        culled = dup == 0;
    }
    ed_args ac = a; /* the culled kernel addresses scene points by their sorted slot */
    ac.sc.sref = s->sref.p;
    ed_brick_filter(m, dist_thres * m->resolution, ac.sc);
    ed_cull cu;
    cu.spos = s->spos.p;
    cu.tiles = s->tiles.p;
    cu.supers = s->supers.p;
    cu.pos_of_orig = d_pos.p;
    cu.tile_bits = s->ed_tilebits.p;
    if (culled) {
        const uint32_t n_super = (s->n_tiles + 63) / 64;
        ED_HIP(hipMemsetAsync(s->ed_tilebits.p, 0, (size_t)n_super * 8, c->stream));
        hipLaunchKernelGGL(k_ed_tile_active, dim3((s->n_tiles + 3) / 4), dim3(256), 0, c->stream, s->spos.p, d_pos.p,
                           s->n_tiles, s->ed_tilebits.p);
    }
    cu.n_tiles = s->n_tiles;
    cu.coord_max = s->coord_max;
    const size_t fused_lds = 2 * (size_t)words * 8;
    const bool fused = culled && fused_lds <= kEdFusedLdsMax && !c->opt.ed_unfused;
    if (fused) { /* bits, walk and score in one kernel, bit arrays in LDS */
        ac.h0 = 0;
        ED_HIP(launch_ed_fused(s, c, n_hyp, fused_lds, ac, cu, o, accept_thresh, d_tests.p));
    }
    for (uint32_t h0 = 0; h0 < n_hyp && !fused; h0 += chunk) {
        const uint32_t nh = std::min(chunk, n_hyp - h0);
        a.h0 = ac.h0 = h0;
        dim3 grid((S + 255) / 256, nh);
        if (culled) {
            ED_HIP(hipMemsetAsync(d_tail.p, 0, (size_t)nh * words * 8, c->stream));
            ED_HIP(hipMemsetAsync(d_corr.p, 0, (size_t)nh * words * 8, c->stream));
            hipLaunchKernelGGL(k_ed_bits_culled, dim3(nh), dim3(256), 0, c->stream, ac, cu);
        } else {
            hipLaunchKernelGGL(k_ed_bits, grid, dim3(256), 0, c->stream, a);
        }
        ED_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_ed_walk, dim3(nh), dim3(64), 0, c->stream, a, o, accept_thresh, d_tests.p);
        ED_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_ed_score, grid, dim3(256), 0, c->stream, a, d_limit.p);
        ED_HIP(hipGetLastError());
    }
    std::vector<unsigned long long> fx(n_hyp);
    ED_HIP(hipMemcpyAsync(counts, d_counts.p, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
    ED_HIP(hipMemcpyAsync(saved, d_saved.p, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
    ED_HIP(hipMemcpyAsync(dropped_at, d_dropped.p, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
    ED_HIP(hipMemcpyAsync(fx.data(), d_fx.p, (size_t)n_hyp * 8, hipMemcpyDeviceToHost, c->stream));
    ED_HIP(hipStreamSynchronize(c->stream));
#undef ED_HIP
    cleanup();
    for (uint32_t h = 0; h < n_hyp; ++h) {
        double raw = (double)fx[h] * (1.0 / kFx);
        /* a dropped pass returns the raw partial sum (:502), a completed one the normalised (:509) */
        scores[h] = dropped_at[h] >= 0 ? raw : raw / (double)m->n;
    }
    return TM_OK;
}

namespace tmx {
}  // namespace tmx

/* The ball mode with everything already on the device: poses at d_poses, the slot (index into uniq) of every
 * hypothesis in s->ed_slot. Leaves counts (s->w_counts), fixed-point sums (s->w_scores_fx), the drop checkpoint
 * (s->ed_dropped) and the saved evaluations (s->ed_saved) on the device, launched on the context's stream. */
static int ed_ball_core(tm_scene* s, tm_model* m, const float* d_poses, uint32_t n_hyp, const uint32_t* uniq,
                        uint32_t n_slots, float radius, float dist_thres, float accept_prob, uint32_t segment = 0,
                        float sufficient = 0.f, uint32_t* d_found = nullptr, const uint32_t* range_starts = nullptr,
                        uint32_t n_ranges = 0, const uint32_t* d_starts_dev = nullptr) {
    /* range_starts (n_ranges + 1 hypothesis offsets, host): independent batches, each with its own word of
     * d_found and its own segments; null = one batch, all hypotheses */
    const uint32_t one_range[2] = {0u, n_hyp};
    if (!range_starts) {
        range_starts = one_range;
        n_ranges = 1;
    }
    tm_ctx* c = s->ctx;
    const uint32_t mwords = (s->n + 63) / 64;
    dbuf<uint32_t>&d_slot = s->ed_slot, &d_cent = s->ed_centres, &d_wprefix = s->ed_wprefix, &d_S = s->ed_S,
                  &d_tests = s->ed_tests, &d_counts = s->w_counts, &d_saved = s->ed_saved, &d_limit = s->ed_limit;
    dbuf<unsigned long long>&d_member = s->ed_member, &d_tail = s->ed_tail, &d_corr = s->ed_corr, &d_fx = s->w_scores_fx;
    dbuf<int32_t>&d_list = s->ed_subset, &d_dropped = s->ed_dropped;
    int rc;
    if ((rc = d_cent.ensure(n_slots)) || (rc = d_member.ensure((size_t)n_slots * mwords)) ||
        (rc = d_wprefix.ensure((size_t)n_slots * mwords)) || (rc = d_S.ensure(n_slots)) ||
        (rc = s->ed_cpos.ensure(n_slots)) || (rc = d_counts.ensure(n_hyp)) ||
        (rc = d_saved.ensure(n_hyp)) || (rc = d_limit.ensure(n_hyp)) || (rc = d_dropped.ensure(n_hyp)) ||
        (rc = d_fx.ensure(n_hyp)) || (rc = d_tests.ensure((size_t)n_slots * 18 + 2)))
        return rc;
    TM_HIP_CHECK(hipMemcpyAsync(d_cent.p, uniq, (size_t)n_slots * 4, hipMemcpyHostToDevice, c->stream));
    /* hypotheses a search step never reaches (later segments after the stop, pending rows after the first waves)
     * keep these: no score, no correspondences, "dropped at once" — never a sufficient hypothesis, never stale
     * values of an earlier call */
    TM_HIP_CHECK(hipMemsetAsync(d_fx.p, 0, (size_t)n_hyp * 8, c->stream));
    TM_HIP_CHECK(hipMemsetAsync(d_counts.p, 0, (size_t)n_hyp * 4, c->stream));
    TM_HIP_CHECK(hipMemsetAsync(d_saved.p, 0, (size_t)n_hyp * 4, c->stream));
    TM_HIP_CHECK(hipMemsetAsync(d_dropped.p, 0, (size_t)n_hyp * 4, c->stream));
    const dim3 grid_n((s->n + 255) / 256, n_slots);
    hipLaunchKernelGGL(k_ball_members, dim3(grid_n.x), dim3(256), 0, c->stream, s->opos.p, s->n, d_cent.p, n_slots,
                       radius * radius, mwords, d_member.p, s->ed_cpos.p);
    hipLaunchKernelGGL(k_ball_prefix, dim3(n_slots), dim3(1024), 0, c->stream, d_member.p, mwords, d_wprefix.p, d_S.p);
    TM_HIP_CHECK(hipGetLastError());
    std::vector<uint32_t> h_S(n_slots);
    TM_HIP_CHECK(hipMemcpyAsync(h_S.data(), d_S.p, (size_t)n_slots * 4, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    uint32_t S_max = 0;
    std::vector<uint32_t> h_tests((size_t)n_slots * 18);
    for (uint32_t k = 0; k < n_slots; ++k) {
        S_max = std::max(S_max, h_S[k]);
        constexpr float step_size = 0.05f; /* reference :422-426 */
        for (uint32_t i = 0; i < 18; ++i) h_tests[(size_t)k * 18 + i] = (uint32_t)(step_size * (i + 1) * h_S[k]);
    }
    const uint32_t list_stride = std::max(1u, S_max);
    TM_FORWARD(d_list.ensure((size_t)n_slots * list_stride));
    TM_HIP_CHECK(hipMemcpyAsync(d_tests.p, h_tests.data(), h_tests.size() * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_ball_list, grid_n, dim3(256), 0, c->stream, d_member.p, d_wprefix.p, s->n, mwords, list_stride,
                       d_list.p);
    TM_HIP_CHECK(hipGetLastError());
    const uint32_t words = std::max(1u, (S_max + 63) / 64);
    /* the first checkpoint's positions of every neighbourhood in curve order (ed_ball::plist) */
    const uint32_t prefix_words = ed_prefix_words(S_max);
    uint32_t B_max = 0;
    std::vector<uint32_t> h_B(n_slots);
    for (uint32_t k = 0; k < n_slots; ++k) {
        const uint32_t S_words = (h_S[k] + 63u) / 64u;
        uint32_t Bw = (h_tests[(size_t)k * 18] + 1024u + 63u) / 64u;
        Bw = std::min(std::min(Bw, S_words), prefix_words);
        h_B[k] = Bw * 64u;
        B_max = std::max(B_max, h_B[k]);
    }
    const bool use_plist = c->opt.ed_prefix_list && B_max > 0;
    /* the whole neighbourhoods the same way for the continuation (ed_ball::blist), while that stays below 1 GiB */
    const bool use_blist = use_plist && c->opt.ed_hist && c->opt.ed_ball_list && (size_t)n_slots * list_stride * 8 <= (1ull << 30);
    if (use_plist) {
        if (use_blist) {
            TM_FORWARD(s->ed_blist.ensure((size_t)n_slots * list_stride));
            TM_FORWARD(s->ed_blist_n.ensure(n_slots));
            TM_HIP_CHECK(hipMemsetAsync(s->ed_blist_n.p, 0, (size_t)n_slots * 4, c->stream));
        }
        TM_FORWARD(s->ed_plist.ensure((size_t)n_slots * B_max));
        TM_FORWARD(s->ed_plist_p.ensure((size_t)n_slots * B_max));
        TM_FORWARD(s->ed_plist_n.ensure(2 * (size_t)n_slots));
        TM_HIP_CHECK(hipMemcpyAsync(s->ed_plist_n.p + n_slots, h_B.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice, c->stream));
        TM_HIP_CHECK(hipMemsetAsync(s->ed_plist_n.p, 0, (size_t)n_slots * 4, c->stream));
        hipLaunchKernelGGL(k_ball_prefix_list, dim3((s->n_sorted + 1023) / 1024, n_slots), dim3(1024), 0, c->stream, s->spos.p,
                           s->n_sorted, d_member.p, d_wprefix.p, mwords, s->ed_plist_n.p + n_slots, B_max, s->ed_plist_n.p,
                           s->ed_plist.p, s->ed_plist_p.p, use_blist ? s->ed_blist_n.p : nullptr,
                           use_blist ? s->ed_blist.p : nullptr, list_stride, s->tiles.p, s->n_tiles, s->ed_cpos.p, radius,
                           1e-5f * radius + 4e-6f * s->coord_max + 1e-12f);
        TM_HIP_CHECK(hipGetLastError());
    }
    const uint64_t budget = 256ull << 20; /* bytes per bit array */
    uint32_t chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_hyp, budget / ((uint64_t)words * 8)));
    chunk = std::min(chunk, 65535u);
    TM_FORWARD(d_tail.ensure((size_t)chunk * words));
    TM_FORWARD(d_corr.ensure((size_t)chunk * words));
    ed_args a;
    a.ball.slot_of_hyp = d_slot.p;
    a.ball.member = d_member.p;
    a.ball.wprefix = d_wprefix.p;
    a.ball.S_of_slot = d_S.p;
    a.ball.tests_of_slot = d_tests.p;
    a.ball.list = d_list.p;
    a.ball.centre_pos = s->ed_cpos.p;
    a.ball.mwords = mwords;
    a.ball.list_stride = list_stride;
    a.ball.radius = radius;
    a.ball.plist = use_plist ? s->ed_plist.p : nullptr;
    a.ball.plist_p = use_plist ? s->ed_plist_p.p : nullptr;
    a.ball.plist_n = s->ed_plist_n.p;
    a.ball.plist_stride = B_max;
    a.ball.blist = use_blist ? s->ed_blist.p : nullptr;
    a.ball.blist_n = s->ed_blist_n.p;
    a.ball.blist_stride = list_stride;
    a.sc.spos = nullptr;
    a.sc.spos3 = nullptr;
    a.sc.sref = s->sref.p; /* the culled kernel addresses scene points by their sorted slot */
    a.sc.mpos = m->pos.p;
    a.sc.mpos3 = m->pos3.p;
    a.sc.mref = m->ref.p;
    a.sc.vox = m->voxels.p;
    a.sc.active = nullptr;
    a.sc.ablate = c->opt.ablate == 30 ? 30 : 0; /* TM_ABLATE=30: one-step evaluation of the fused kernel (timing comparison) */
    a.sc.kind = m->metric_kind;
    a.sc.radius = m->metric_radius;
    a.sc.dmax = dist_thres * m->resolution;
    a.sc.bcode = nullptr;
    a.sc.bcode_tile = nullptr;
    a.sc.bx = m->bdim[0];
    a.sc.by = m->bdim[1];
    a.sc.code_max = 255u;
    a.sc.poses = d_poses;
    a.sc.counts = d_counts.p;
    a.sc.scores_fx = d_fx.p;
    a.sc.stat_slots = nullptr;
    a.sc.g = m->grid;
    a.sc.d2max = sq_threshold(dist_thres * m->resolution);
    a.sc.n_hyp = n_hyp;
    a.sc.hyp_per_block = 1;
    a.opos = s->opos.p;
    a.otan = s->otan.p;
    a.subset = nullptr;
    a.S = S_max;
    a.words = words;
    a.tail_bits = d_tail.p;
    a.corr_bits = d_corr.p;
    a.stop = nullptr;
    a.rng_starts = nullptr;
    a.rng_stop = nullptr;
    a.seg_len = a.seg_step = 0;
    a.row_info = nullptr;
    a.ord_lo = 0;
    a.ord_hi = 0xFFFFFFFFu;
    a.prefix_words = ed_prefix_words(a.S);
    ed_brick_filter(m, dist_thres * m->resolution, a.sc);
    TM_FORWARD(ed_super_filter(m, a.sc));
    ed_args as = a; /* k_ed_score addresses scene points by original index */
    as.sc.sref = s->oref.p;
    ed_cull cu;
    cu.tile_bits = nullptr;
    cu.spos = s->spos.p;
    cu.tiles = s->tiles.p;
    cu.supers = s->supers.p;
    cu.pos_of_orig = nullptr;
    cu.n_tiles = s->n_tiles;
    cu.coord_max = s->coord_max;
    ed_out o;
    o.counts = d_counts.p;
    o.dropped_at = d_dropped.p;
    o.saved = d_saved.p;
    o.limit = d_limit.p;
    const float accept_thresh = accept_prob * (float)m->n;
    const size_t fused_lds = 2 * (size_t)words * 8;
    /* neighbourhoods beyond 196k points do not fit the full kernel's LDS bit arrays; the prefix kernel and the
     * wide continuation (bit rows in global memory) do not need them, so batches small enough for the wide
     * continuation keep the two-step rule at any neighbourhood size */
    const bool big_lds = fused_lds > kEdFusedLdsMax;
    const bool fused = (!big_lds || n_hyp <= 4096u || (segment && segment <= 4096u && d_found)) &&
                       !c->opt.ed_unfused && !(big_lds && (c->opt.ed_narrow || c->opt.ed_one_kernel));
    if (fused && segment && d_found) {
        /* tm_search_batch: only the first sufficient hypothesis in visit order is wanted (reference :330-338: later
         * ones are scored without effect). The hypotheses go through in segments; after each, *d_found takes the
         * first sufficient index of the segment, and the kernels of all later segments leave at once. */
        if (n_ranges > 1 && d_starts_dev && !c->opt.ed_one_kernel && !c->opt.ed_narrow && !c->opt.search_per_range) {
            /* The batches are independent (the chunks of one find_parallel round): segment step k of ALL of them
             * goes through in one set of launches — workgroup (x, b) of the prefix kernel takes the x-th hypothesis of
             * batch b's k-th segment, the pending hypotheses of all batches share the wide continuation, and one
             * workgroup per batch writes its stop word. A round of seven chunks costs max(segments) steps of six
             * launches instead of sum(segments) x 7, and the latency-bound continuation kernels (a chain of
             * dependent gathers per pending row) run once per step instead of once per batch. */
            uint32_t steps = 0;
            for (uint32_t b = 0; b < n_ranges; ++b)
                steps = std::max(steps, (range_starts[b + 1] - range_starts[b] + segment - 1) / segment);
            const size_t pend_cap = (size_t)n_ranges * segment;
            TM_FORWARD(s->ed_pending.ensure(pend_cap + steps + 1));
            uint32_t* d_n_steps = s->ed_pending.p + pend_cap;
            TM_HIP_CHECK(hipMemsetAsync(d_n_steps, 0, (size_t)steps * 4, c->stream));
            a.h0 = 0;
            a.stop = nullptr;
            a.rng_starts = d_starts_dev;
            a.rng_stop = d_found;
            a.seg_len = segment;
            const uint32_t rows_cap = (uint32_t)std::min<size_t>(chunk, pend_cap);
            for (uint32_t k = 0; k < steps; ++k) {
                a.seg_step = k;
                ed_seg_check chk;
                chk.counts = d_counts.p;
                chk.fx = d_fx.p;
                chk.sufficient = sufficient;
                chk.found = d_found;
                chk.starts = d_starts_dev;
                chk.n_ranges = n_ranges;
                chk.seg_len = segment;
                chk.step = k;
                TM_HIP_CHECK(launch_ed_fused(s, c, (uint32_t)pend_cap, fused_lds, a, cu, o, accept_thresh, d_tests.p, rows_cap,
                                             n_ranges, d_n_steps + k, &chk));
                /* Every batch that has its sufficient hypothesis makes the kernels of its later steps leave at once —
                 * but a step is still ten launches. After the first step (the one that usually settles a round with
                 * matches in it) and after every third one the batches' stop words come back (4 bytes each, one wait):
                 * when all are set the remaining steps are not queued at all. */
                if (c->opt.search_step_check && k + 1 < steps && (k == 0 || k % 3 == 0)) {
                    std::vector<uint32_t> h_found(n_ranges);
                    TM_HIP_CHECK(hipMemcpyAsync(h_found.data(), d_found, (size_t)n_ranges * 4, hipMemcpyDeviceToHost, c->stream));
                    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
                    bool all = true;
                    for (uint32_t b = 0; b < n_ranges && all; ++b)
                        all = h_found[b] != 0xFFFFFFFFu || range_starts[b + 1] - range_starts[b] <= (k + 1) * segment;
                    if (all) break;
                }
            }
            a.rng_starts = nullptr;
        } else
        for (uint32_t b = 0; b < n_ranges; ++b) {
            a.stop = d_found + b;
            for (uint32_t h0 = range_starts[b]; h0 < range_starts[b + 1]; h0 += segment) {
                const uint32_t nh = std::min(segment, range_starts[b + 1] - h0);
                a.h0 = h0;
                ed_seg_check chk;
                chk.counts = d_counts.p;
                chk.fx = d_fx.p;
                chk.sufficient = sufficient;
                chk.found = d_found + b;
                chk.h0 = h0;
                chk.h1 = h0 + nh;
                TM_HIP_CHECK(launch_ed_fused(s, c, nh, fused_lds, a, cu, o, accept_thresh, d_tests.p, std::min(chunk, nh), 0,
                                             nullptr, &chk));
            }
        }
    } else if (fused) {
        /* the whole batch at once: with ~2 % of 65,536 hypotheses pending the wide continuation has over a
         * thousand rows and loses to one workgroup per pending hypothesis (21.4 against 18.0 ms per batch); it
         * pays for the few rows of a segment (end-to-end search 0.030 against 0.034 s) */
        a.h0 = 0;
        TM_HIP_CHECK(launch_ed_fused(s, c, n_hyp, fused_lds, a, cu, o, accept_thresh, d_tests.p,
                                     n_hyp <= 4096u || big_lds ? chunk : 0u));
        if (c->opt.debug_ed) { /* how many hypotheses the prefix kernel could not decide */
            uint32_t n_pend = 0;
            TM_HIP_CHECK(hipMemcpyAsync(&n_pend, s->ed_pending.p + n_hyp, 4, hipMemcpyDeviceToHost, c->stream));
            TM_HIP_CHECK(hipStreamSynchronize(c->stream));
            fprintf(stderr, "[early drop] %u hypotheses, %u slots, largest neighbourhood %u (%u words), %u pending\n", n_hyp,
                    n_slots, S_max, words, n_pend);
        }
    }
    for (uint32_t h0 = 0; h0 < n_hyp && !fused; h0 += chunk) {
        const uint32_t nh = std::min(chunk, n_hyp - h0);
        a.h0 = as.h0 = h0;
        TM_HIP_CHECK(hipMemsetAsync(d_tail.p, 0, (size_t)nh * words * 8, c->stream));
        TM_HIP_CHECK(hipMemsetAsync(d_corr.p, 0, (size_t)nh * words * 8, c->stream));
        hipLaunchKernelGGL(k_ed_bits_culled, dim3(nh), dim3(256), 0, c->stream, a, cu);
        hipLaunchKernelGGL(k_ed_walk, dim3(nh), dim3(64), 0, c->stream, a, o, accept_thresh, d_tests.p);
        hipLaunchKernelGGL(k_ed_score, dim3((S_max + 255) / 256, nh), dim3(256), 0, c->stream, as, d_limit.p);
        TM_HIP_CHECK(hipGetLastError());
    }
    if (d_found && !(fused && segment)) { /* no segments were used: one look over every batch */
        for (uint32_t b = 0; b < n_ranges; ++b)
            hipLaunchKernelGGL(k_segment_first, dim3(1), dim3(1024), 0, c->stream, d_counts.p, d_fx.p, range_starts[b],
                               range_starts[b + 1], sufficient, d_found + b);
        TM_HIP_CHECK(hipGetLastError());
    }
    return TM_OK;
}

/* distinct centres in ascending order and each hypothesis' slot among them; hypotheses come in runs of one centre
 * (a few dozen distinct ones in 65,536), so a look-up per run and a sort of the distinct values replaces a sort of
 * all hypotheses (1.5 ms of host time per batch) */
static bool ed_slots_of_centres(const uint32_t* centers, uint32_t n_hyp, uint32_t n_scene, std::vector<uint32_t>& uniq,
                                std::vector<uint32_t>& slot_of_hyp) {
    std::unordered_map<uint32_t, uint32_t> id;
    uint32_t prev = 0, prev_id = 0;
    for (uint32_t h = 0; h < n_hyp; ++h) {
        const uint32_t cidx = centers[h];
        if (cidx >= n_scene) return false;
        if (h == 0 || cidx != prev) {
            auto it = id.find(cidx);
            if (it == id.end()) {
                it = id.emplace(cidx, (uint32_t)uniq.size()).first;
                uniq.push_back(cidx);
            }
            prev = cidx;
            prev_id = it->second;
        }
        slot_of_hyp[h] = prev_id;
    }
    std::vector<uint32_t> order(uniq.size()), remap(uniq.size());
    for (uint32_t k = 0; k < order.size(); ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return uniq[x] < uniq[y]; });
    std::vector<uint32_t> sorted(uniq.size());
    for (uint32_t k = 0; k < order.size(); ++k) {
        remap[order[k]] = k;
        sorted[k] = uniq[order[k]];
    }
    uniq.swap(sorted);
    for (uint32_t h = 0; h < n_hyp; ++h) slot_of_hyp[h] = remap[slot_of_hyp[h]];
    return true;
}

extern "C" int tm_score_hypotheses_early_drop_ball(tm_scene* s, tm_model* m, const float* poses16, uint32_t n_hyp,
                                                   const uint32_t* centers, float radius, float dist_thres,
                                                   float accept_prob, uint32_t* counts, double* scores,
                                                   int32_t* dropped_at, uint32_t* saved) {
    TM_GUARD(s);
    TM_REQUIRE(s && m, TM_ERR_INVALID, "null scene or model");
    TM_REQUIRE(s->ctx == m->ctx, TM_ERR_INVALID, "scene and model belong to different contexts");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    TM_REQUIRE(counts && scores && dropped_at && saved, TM_ERR_INVALID, "null outputs");
    if (n_hyp == 0) return TM_OK;
    TM_REQUIRE(poses16 && centers, TM_ERR_INVALID, "null poses or centres");
    TM_REQUIRE(radius >= 0.f, TM_ERR_INVALID, "negative radius");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    /* distinct centres -> slots */
    std::vector<uint32_t> uniq, slot_of_hyp(n_hyp);
    TM_REQUIRE(ed_slots_of_centres(centers, n_hyp, s->n, uniq, slot_of_hyp), TM_ERR_INVALID, "centre index out of range");
    TM_FORWARD(s->ed_slot.ensure(n_hyp));
    TM_FORWARD(s->w_poses.ensure((size_t)n_hyp * 16));
    TM_HIP_CHECK(hipMemcpyAsync(s->ed_slot.p, slot_of_hyp.data(), (size_t)n_hyp * 4, hipMemcpyHostToDevice, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(s->w_poses.p, poses16, (size_t)n_hyp * 64, hipMemcpyHostToDevice, c->stream));
    TM_FORWARD(ed_ball_core(s, m, s->w_poses.p, n_hyp, uniq.data(), (uint32_t)uniq.size(), radius, dist_thres,
                            accept_prob));
    dbuf<uint32_t>&d_counts = s->w_counts, &d_saved = s->ed_saved;
    dbuf<int32_t>& d_dropped = s->ed_dropped;
    dbuf<unsigned long long>& d_fx = s->w_scores_fx;
    std::vector<unsigned long long> fx(n_hyp);
    TM_HIP_CHECK(hipMemcpyAsync(counts, d_counts.p, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(saved, d_saved.p, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(dropped_at, d_dropped.p, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(fx.data(), d_fx.p, (size_t)n_hyp * 8, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (uint32_t h = 0; h < n_hyp; ++h) {
        double raw = (double)fx[h] * (1.0 / kFx);
        scores[h] = dropped_at[h] >= 0 ? raw : raw / (double)m->n; /* :502 / :509 */
    }
    return TM_OK;
}

namespace tmx {
/* scores as project_ returns them: the raw partial sum for a dropped hypothesis (:502), sum / |model| otherwise (:509) */
__global__ void k_ed_scores(const unsigned long long* __restrict__ fx, const int32_t* __restrict__ dropped, uint32_t n,
                            double inv_scale, double n_model, double* __restrict__ out) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n) return;
    const double raw = (double)fx[h] * inv_scale;
    out[h] = dropped[h] >= 0 ? raw : raw / n_model;
}
}  // namespace tmx

/* The same with the poses already on the device and the results left there (pipelines, bench): only the centre
 * indices (4 bytes per hypothesis) come from the host — their distinct values name the neighbourhoods to build. */
extern "C" int tm_score_hypotheses_early_drop_ball_dev(tm_scene* s, tm_model* m, const float* d_poses16, uint32_t n_hyp,
                                                       const uint32_t* centers, float radius, float dist_thres,
                                                       float accept_prob, uint32_t* d_counts, double* d_scores,
                                                       int32_t* d_dropped_at, uint32_t* d_saved) {
    TM_GUARD(s);
    TM_REQUIRE(s && m, TM_ERR_INVALID, "null scene or model");
    TM_REQUIRE(s->ctx == m->ctx, TM_ERR_INVALID, "scene and model belong to different contexts");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    if (n_hyp == 0) return TM_OK;
    TM_REQUIRE(d_poses16 && centers && d_counts && d_scores && d_dropped_at && d_saved, TM_ERR_INVALID,
               "tm_score_hypotheses_early_drop_ball_dev: null argument");
    TM_REQUIRE(radius >= 0.f, TM_ERR_INVALID, "negative radius");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    std::vector<uint32_t> uniq, slot_of_hyp(n_hyp);
    TM_REQUIRE(ed_slots_of_centres(centers, n_hyp, s->n, uniq, slot_of_hyp), TM_ERR_INVALID, "centre index out of range");
    TM_FORWARD(s->ed_slot.ensure(n_hyp));
    TM_HIP_CHECK(hipMemcpyAsync(s->ed_slot.p, slot_of_hyp.data(), (size_t)n_hyp * 4, hipMemcpyHostToDevice, c->stream));
    TM_FORWARD(ed_ball_core(s, m, d_poses16, n_hyp, uniq.data(), (uint32_t)uniq.size(), radius, dist_thres, accept_prob));
    TM_HIP_CHECK(hipMemcpyAsync(d_counts, s->w_counts.p, (size_t)n_hyp * 4, hipMemcpyDeviceToDevice, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(d_saved, s->ed_saved.p, (size_t)n_hyp * 4, hipMemcpyDeviceToDevice, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(d_dropped_at, s->ed_dropped.p, (size_t)n_hyp * 4, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_ed_scores, dim3((n_hyp + 255) / 256), dim3(256), 0, c->stream, s->w_scores_fx.p, s->ed_dropped.p,
                       n_hyp, 1.0 / kFx, (double)m->n, d_scores);
    TM_HIP_CHECK(hipGetLastError());
    return TM_OK;
}

namespace tmx {

/* hypothesis -> slot of its pair's first point (provenance word 0 = pair index) */
__global__ void k_batch_slots(const uint4* __restrict__ prov, uint32_t n_hyp, const uint32_t* __restrict__ of_pair,
                              uint32_t* __restrict__ of_hyp) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < n_hyp) of_hyp[h] = of_pair[prov[h].x];
}

/* first hypothesis in visit order with score > 0 and (float)count > sufficient (reference :330-338); one block.
 * fx != null: early-drop results (score = sum / 2^40, divided by n_model when the scan ran to the end);
 * else scores holds the full-pass scores. */
__global__ void __launch_bounds__(1024)
k_first_sufficient(const uint32_t* __restrict__ counts, const unsigned long long* __restrict__ fx,
                   const double* __restrict__ scores, const int32_t* __restrict__ dropped,
                   const uint32_t* __restrict__ starts, float sufficient, double n_model, const float* __restrict__ poses,
                   tm_batch_result* out, const uint32_t* __restrict__ found = nullptr) {
    /* workgroup b = batch b = hypotheses [starts[b], starts[b + 1]) of the arrays; the record's index counts from the
     * batch's first hypothesis (one launch for the batches of a round: seven launches per round before) */
    const uint32_t h0 = starts[blockIdx.x], n_hyp = starts[blockIdx.x + 1] - h0;
    out += blockIdx.x;
    if (found) found += blockIdx.x;
    __shared__ uint32_t s_first;
    if (threadIdx.x == 0) s_first = found ? *found : 0xFFFFFFFFu; /* found: the segments have already looked */
    __syncthreads();
    uint32_t first = 0xFFFFFFFFu;
    for (uint32_t h = h0 + threadIdx.x; h < (found ? 0u : h0 + n_hyp); h += blockDim.x) {
        const bool positive = fx ? fx[h] != 0ull : scores[h] > 0.0;
        if (positive && (float)counts[h] > sufficient) {
            first = h;
            break; /* this thread's later candidates come later in visit order */
        }
    }
    if (first != 0xFFFFFFFFu) atomicMin(&s_first, first);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t h = s_first;
        out->index = h == 0xFFFFFFFFu ? -1 : (int32_t)(h - h0);
        out->n_hypotheses = n_hyp;
        out->count = 0;
        out->dropped_at = -1;
        out->score = 0.0;
        if (h != 0xFFFFFFFFu) {
            out->count = counts[h];
            if (fx) {
                const double raw = (double)fx[h] * (1.0 / kFx);
                out->dropped_at = dropped[h];
                out->score = dropped[h] >= 0 ? raw : raw / n_model;
            } else {
                out->score = scores[h];
            }
        }
    }
    if (s_first != 0xFFFFFFFFu && threadIdx.x < 16) out->pose[threadIdx.x] = poses[(size_t)s_first * 16 + threadIdx.x];
}

}  // namespace tmx

namespace tmx {
/* first hypothesis of every batch: provenance word 0 = pair index, ascending; starts[b] = first hypothesis whose
 * pair is >= pair_offsets[b] */
__global__ void k_batch_starts(const uint4* __restrict__ prov, uint32_t n_hyp, const uint32_t* __restrict__ pair_offsets,
                               uint32_t n_batches, uint32_t* __restrict__ starts) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_batches) return;
    const uint32_t want = pair_offsets[b];
    uint32_t lo = 0, hi = n_hyp;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (prov[mid].x < want) lo = mid + 1; else hi = mid;
    }
    starts[b] = lo;
}
}  // namespace tmx

extern "C" int tm_search_batches(tm_scene* s, tm_model* m, const uint32_t* ij_pairs, uint32_t n_pairs,
                                 const uint32_t* batch_offsets, uint32_t n_batches, float minf, float maxf, int force_up,
                                 int query_limit, float radius, float dist_thres, float accept_prob, int early_drop,
                                 float sufficient, tm_batch_result* out) {
    TM_GUARD(s);
    TM_REQUIRE(s && m && out && batch_offsets, TM_ERR_INVALID, "tm_search_batches: null argument");
    TM_REQUIRE(s->ctx == m->ctx, TM_ERR_INVALID, "scene and model belong to different contexts");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    TM_REQUIRE(query_limit > 0, TM_ERR_INVALID, "tm_search_batches: query_limit must be positive");
    TM_REQUIRE(radius >= 0.f, TM_ERR_INVALID, "negative radius");
    TM_REQUIRE(n_batches >= 1 && batch_offsets[0] == 0 && batch_offsets[n_batches] == n_pairs, TM_ERR_INVALID,
               "tm_search_batches: batch offsets must run from 0 to n_pairs");
    for (uint32_t b = 0; b < n_batches; ++b)
        TM_REQUIRE(batch_offsets[b] <= batch_offsets[b + 1], TM_ERR_INVALID, "tm_search_batches: batch offsets must ascend");
    for (uint32_t b = 0; b < n_batches; ++b) {
        memset(&out[b], 0, sizeof(out[b]));
        out[b].index = -1;
        out[b].dropped_at = -1;
    }
    if (n_pairs == 0) return m->pairs_ready ? TM_OK : TM_ERR_NOT_INITIALIZED;
    TM_REQUIRE(ij_pairs, TM_ERR_INVALID, "tm_search_batches: null pairs");
    TM_REQUIRE((uint64_t)n_pairs * (uint64_t)query_limit <= 0x7FFFFFFFull, TM_ERR_INVALID, "tm_search_batches: too many pairs");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    /* distinct first points -> slots (ascending, as the per-hypothesis entry point numbers them) */
    std::vector<uint32_t> uniq, slot_of_pair(n_pairs);
    for (uint32_t p = 0; p < n_pairs; ++p) {
        TM_REQUIRE(ij_pairs[2 * p] < s->n && ij_pairs[2 * p + 1] < s->n, TM_ERR_INVALID, "tm_search_batches: index out of range");
        uniq.push_back(ij_pairs[2 * p]);
    }
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    for (uint32_t p = 0; p < n_pairs; ++p)
        slot_of_pair[p] = early_drop ? (uint32_t)(std::lower_bound(uniq.begin(), uniq.end(), ij_pairs[2 * p]) - uniq.begin())
                                     : ij_pairs[2 * p]; /* full pass: the centre point itself */
    const uint32_t cap = n_pairs * (uint32_t)query_limit;
    TM_FORWARD(s->sb_ij.ensure((size_t)n_pairs * 2));
    TM_FORWARD(s->sb_pair_slot.ensure((size_t)n_pairs + n_batches + 1)); /* + the batches' pair offsets */
    TM_FORWARD(s->sb_prov.ensure(cap));
    TM_FORWARD(s->sb_result.ensure(n_batches));
    TM_FORWARD(s->sb_centres.ensure((size_t)cap + 2 * (size_t)n_batches + 2)); /* centres | found words | batch starts */
    TM_FORWARD(s->w_poses.ensure((size_t)cap * 16));
    TM_HIP_CHECK(hipMemcpyAsync(s->sb_ij.p, ij_pairs, (size_t)n_pairs * 8, hipMemcpyHostToDevice, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(s->sb_pair_slot.p, slot_of_pair.data(), (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(s->sb_pair_slot.p + n_pairs, batch_offsets, ((size_t)n_batches + 1) * 4, hipMemcpyHostToDevice,
                                c->stream));
    uint32_t n_hyp = 0, total = 0;
    TM_FORWARD(tm_generate_hypotheses_dev(s, m, s->sb_ij.p, n_pairs, minf, maxf, force_up, query_limit, s->w_poses.p,
                                          (uint32_t*)s->sb_prov.p, cap, &n_hyp, &total));
    if (n_hyp == 0) return TM_OK;
    uint32_t* d_found = s->sb_centres.p + cap;
    uint32_t* d_starts = d_found + n_batches + 1;
    std::vector<uint32_t> starts((size_t)n_batches + 1);
    hipLaunchKernelGGL(k_batch_starts, dim3((n_batches + 256) / 256), dim3(256), 0, c->stream, s->sb_prov.p, n_hyp,
                       s->sb_pair_slot.p + n_pairs, n_batches, d_starts);
    TM_HIP_CHECK(hipGetLastError());
    TM_HIP_CHECK(hipMemcpyAsync(starts.data(), d_starts, starts.size() * 4, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipMemsetAsync(d_found, 0xFF, (size_t)n_batches * 4, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (early_drop) {
        TM_FORWARD(s->ed_slot.ensure(n_hyp));
        hipLaunchKernelGGL(k_batch_slots, dim3((n_hyp + 255) / 256), dim3(256), 0, c->stream, s->sb_prov.p, n_hyp,
                           s->sb_pair_slot.p, s->ed_slot.p);
        TM_HIP_CHECK(hipGetLastError());
        /* segments of 2048 hypotheses (one resident round of the prefix kernel); TM_SEARCH_SEGMENT=0: all at once */
        const uint32_t segment = s->ctx->opt.search_segment >= 0 ? (uint32_t)s->ctx->opt.search_segment : 2048u;
        TM_FORWARD(ed_ball_core(s, m, s->w_poses.p, n_hyp, uniq.data(), (uint32_t)uniq.size(), radius, dist_thres,
                                accept_prob, segment, sufficient, d_found, starts.data(), n_batches, d_starts));
        hipLaunchKernelGGL(k_first_sufficient, dim3(n_batches), dim3(1024), 0, c->stream, s->w_counts.p, s->w_scores_fx.p,
                           (const double*)nullptr, s->ed_dropped.p, (const uint32_t*)d_starts, sufficient, (double)m->n,
                           s->w_poses.p, s->sb_result.p, (const uint32_t*)d_found);
    } else {
        /* full pass: centre of a hypothesis = first point of its pair */
        TM_FORWARD(s->w_counts.ensure(n_hyp));
        TM_FORWARD(s->w_scores.ensure(n_hyp));
        hipLaunchKernelGGL(k_batch_slots, dim3((n_hyp + 255) / 256), dim3(256), 0, c->stream, s->sb_prov.p, n_hyp,
                           s->sb_pair_slot.p, s->sb_centres.p);
        TM_HIP_CHECK(hipGetLastError());
        TM_FORWARD(tm_score_hypotheses_ball_dev(s, m, s->w_poses.p, n_hyp, s->sb_centres.p, radius, dist_thres,
                                                s->w_counts.p, s->w_scores.p));
        hipLaunchKernelGGL(k_first_sufficient, dim3(n_batches), dim3(1024), 0, c->stream, s->w_counts.p,
                           (const unsigned long long*)nullptr, s->w_scores.p, (const int32_t*)nullptr,
                           (const uint32_t*)d_starts, sufficient, (double)m->n, s->w_poses.p, s->sb_result.p,
                           (const uint32_t*)nullptr);
    }
    TM_HIP_CHECK(hipGetLastError());
    TM_HIP_CHECK(hipMemcpyAsync(out, s->sb_result.p, sizeof(*out) * n_batches, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return TM_OK;
}

extern "C" int tm_search_batch(tm_scene* s, tm_model* m, const uint32_t* ij_pairs, uint32_t n_pairs, float minf,
                               float maxf, int force_up, int query_limit, float radius, float dist_thres,
                               float accept_prob, int early_drop, float sufficient, tm_batch_result* out) {
    TM_GUARD(s);
    TM_REQUIRE(out, TM_ERR_INVALID, "tm_search_batch: null argument");
    const uint32_t offsets[2] = {0u, n_pairs};
    return tm_search_batches(s, m, ij_pairs, n_pairs, offsets, 1, minf, maxf, force_up, query_limit, radius, dist_thres,
                             accept_prob, early_drop, sufficient, out);
}

namespace tmx {
/* project_ (full pass) of every pose on a short list of scene points: one thread per (pose, listed point), straight
 * from the arrays in original order — no walk over the scene's tiles, which is what a whole-scene kernel spends
 * its time on when the list is a few thousand points */
__global__ __launch_bounds__(256) void k_score_points(score_args sc /* sref = original order */,
                                                      const float4* __restrict__ opos, const float4* __restrict__ otan,
                                                      const int32_t* __restrict__ points, uint32_t n_points) {
    const uint32_t pos = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
    long long fx = 0;
    uint32_t cnt = 0;
    if (pos < n_points) {
        const uint32_t idx = (uint32_t)points[pos];
        if (otan[idx].w == 0.f) { /* mask_ (:434) */
            const float4 P = opos[idx];
            const pose12 T = load_pose(sc.poses + (size_t)h * 16);
            uint32_t pass = 0;
            cnt = eval_point(sc, T, make3(P.x, P.y, P.z), idx, fx, pass, nullptr, nullptr);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        fx += __shfl_down(fx, off);
        cnt += __shfl_down(cnt, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt) atomicAdd(&sc.counts[h], cnt);
        if (fx) atomicAdd(&sc.scores_fx[h], (unsigned long long)fx);
    }
}
}  // namespace tmx

extern "C" int tm_score_hypotheses_on_points(tm_scene* s, tm_model* m, const float* poses16, uint32_t n_hyp,
                                             const int32_t* points, uint32_t n_points, float dist_thres,
                                             uint32_t* counts, double* scores) {
    TM_GUARD(s);
    TM_REQUIRE(s && m, TM_ERR_INVALID, "null scene or model");
    TM_REQUIRE(s->ctx == m->ctx, TM_ERR_INVALID, "scene and model belong to different contexts");
    TM_REQUIRE(m->voxels_ready, TM_ERR_NOT_INITIALIZED, "voxel grid not built");
    TM_REQUIRE(counts && scores, TM_ERR_INVALID, "null outputs");
    if (n_hyp == 0) return TM_OK;
    TM_REQUIRE(poses16 && (points || n_points == 0), TM_ERR_INVALID, "null inputs");
    TM_REQUIRE(n_hyp <= 65535u, TM_ERR_INVALID, "tm_score_hypotheses_on_points: at most 65,535 poses per call");
    for (uint32_t h = 0; h < n_hyp; ++h) {
        counts[h] = 0;
        scores[h] = 0.0;
    }
    if (n_points == 0) return TM_OK;
    /* every point at most once (a set, like the subset of tm_score_hypotheses) */
    if (s->h_act.size() != s->n) s->h_act.assign(s->n, 0);
    bool bad = false;
    uint32_t k_done = 0;
    for (; k_done < n_points; ++k_done) {
        const int32_t i = points[k_done];
        if (i < 0 || (uint32_t)i >= s->n || s->h_act[(uint32_t)i]) {
            bad = true;
            break;
        }
        s->h_act[(uint32_t)i] = 1;
    }
    for (uint32_t k = 0; k < k_done; ++k) s->h_act[(uint32_t)points[k]] = 0;
    TM_REQUIRE(!bad, TM_ERR_INVALID, "tm_score_hypotheses_on_points: index out of range or listed twice");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_FORWARD(s->w_poses.ensure((size_t)n_hyp * 16));
    TM_FORWARD(s->w_counts.ensure(n_hyp));
    TM_FORWARD(s->w_scores_fx.ensure(n_hyp));
    TM_FORWARD(s->ed_subset.ensure(n_points));
    TM_HIP_CHECK(hipMemcpyAsync(s->w_poses.p, poses16, (size_t)n_hyp * 64, hipMemcpyHostToDevice, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(s->ed_subset.p, points, (size_t)n_points * 4, hipMemcpyHostToDevice, c->stream));
    TM_HIP_CHECK(hipMemsetAsync(s->w_counts.p, 0, (size_t)n_hyp * 4, c->stream));
    TM_HIP_CHECK(hipMemsetAsync(s->w_scores_fx.p, 0, (size_t)n_hyp * 8, c->stream));
    score_args sc;
    sc.spos = nullptr;
    sc.spos3 = nullptr;
    sc.sref = s->oref.p;
    sc.mpos = m->pos.p;
    sc.mpos3 = m->pos3.p;
    sc.mref = m->ref.p;
    sc.vox = m->voxels.p;
    sc.active = nullptr;
    sc.ablate = 0;
    sc.kind = m->metric_kind;
    sc.radius = m->metric_radius;
    sc.dmax = dist_thres * m->resolution;
    sc.bx = m->bdim[0];
    sc.by = m->bdim[1];
    sc.poses = s->w_poses.p;
    sc.counts = s->w_counts.p;
    sc.scores_fx = s->w_scores_fx.p;
    sc.stat_slots = nullptr;
    sc.g = m->grid;
    sc.d2max = sq_threshold(dist_thres * m->resolution);
    sc.n_hyp = n_hyp;
    sc.hyp_per_block = 1;
    ed_brick_filter(m, dist_thres * m->resolution, sc);
    hipLaunchKernelGGL(k_score_points, dim3((n_points + 255) / 256, n_hyp), dim3(256), 0, c->stream, sc, s->opos.p,
                       s->otan.p, s->ed_subset.p, n_points);
    TM_HIP_CHECK(hipGetLastError());
    std::vector<unsigned long long> fx(n_hyp);
    TM_HIP_CHECK(hipMemcpyAsync(counts, s->w_counts.p, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipMemcpyAsync(fx.data(), s->w_scores_fx.p, (size_t)n_hyp * 8, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (uint32_t h = 0; h < n_hyp; ++h) scores[h] = (double)fx[h] * (1.0 / kFx) / (double)m->n; /* :509 */
    return TM_OK;
}

/* radius_search_inclusive for several centres with the index lists built on the device: membership bits,
 * per-word prefix counts and an ordered scatter (the kernels of the ball mode above); only counts[k] indices
 * per centre travel back. */
/* What the sampling loop of find_in_subset needs of nn = radius_search_inclusive(radius, centre) (reference
 * include/impl/scene.hpp:273-302): its size (inner_bound :276-281, and the range of the shuffle's draws :285) and its
 * members with tangent_mask_ set — every other member fails the first filter of :290 whatever its index is. So only
 * those travel: counts[k] = size of nn_k, t_counts[k] = its tangent members, and their (rank, index) pairs, slot after
 * slot, each slot in ascending index order; rank = position in nn_k's ascending order (the order
 * tm_scene_radius_lists hands out). A neighbourhood of 60,000 points with 4 % tangent points is 19 KB instead of
 * 240 KB. *n_pairs = total; TM_ERR_CAPACITY (counts, t_counts and *n_pairs valid) when cap_pairs is too small. */
extern "C" int tm_scene_radius_tangent_lists(tm_scene* s, const uint32_t* centers, uint32_t n_centers, float radius,
                                             uint32_t* counts, uint32_t* t_counts, uint32_t* pairs, uint32_t cap_pairs,
                                             uint32_t* n_pairs) {
    TM_GUARD(s);
    TM_REQUIRE(s && counts && t_counts && n_pairs && (centers || n_centers == 0), TM_ERR_INVALID,
               "tm_scene_radius_tangent_lists: null argument");
    *n_pairs = 0;
    if (n_centers == 0) return TM_OK;
    for (uint32_t k = 0; k < n_centers; ++k)
        TM_REQUIRE(centers[k] < s->n, TM_ERR_INVALID, "tm_scene_radius_tangent_lists: centre index out of range");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    TM_FORWARD(scene_tangent_index(s));
    const uint32_t mwords = (s->n + 63) / 64;
    TM_FORWARD(s->ed_centres.ensure(n_centers));
    TM_FORWARD(s->ed_member.ensure((size_t)n_centers * mwords));
    TM_FORWARD(s->ed_wprefix.ensure((size_t)n_centers * mwords));
    TM_FORWARD(s->ed_S.ensure(2 * (size_t)n_centers)); /* sizes, then tangent counts */
    TM_FORWARD(s->ed_cpos.ensure(n_centers));
    const uint32_t stride = std::max(1u, s->n_tangent);
    TM_FORWARD(s->ed_tlist.ensure(4 * (size_t)n_centers * stride)); /* staging + CSR, 2 words per entry each */
    uint2* d_staging = reinterpret_cast<uint2*>(s->ed_tlist.p);
    uint2* d_csr = d_staging + (size_t)n_centers * stride;
    TM_HIP_CHECK(hipMemcpyAsync(s->ed_centres.p, centers, (size_t)n_centers * 4, hipMemcpyHostToDevice, c->stream));
    const dim3 grid_n((s->n + 255) / 256, n_centers);
    hipLaunchKernelGGL(k_ball_members, dim3(grid_n.x), dim3(256), 0, c->stream, s->opos.p, s->n, s->ed_centres.p, n_centers,
                       radius * radius, mwords, s->ed_member.p, s->ed_cpos.p);
    hipLaunchKernelGGL(k_ball_prefix, dim3(n_centers), dim3(1024), 0, c->stream, s->ed_member.p, mwords,
                       s->ed_wprefix.p, s->ed_S.p);
    hipLaunchKernelGGL(k_ball_tangent, dim3(n_centers), dim3(1024), 0, c->stream, s->ed_member.p, s->ed_wprefix.p, mwords,
                       s->d_tangent_idx.p, s->n_tangent, stride, d_staging, s->ed_S.p + n_centers);
    hipLaunchKernelGGL(k_ball_tangent_compact, dim3((stride + 255) / 256, n_centers), dim3(256), 0, c->stream, d_staging,
                       stride, s->ed_S.p + n_centers, d_csr);
    TM_HIP_CHECK(hipGetLastError());
    /* sizes and tangent counts in one read-back, then exactly the pairs that were written */
    const size_t head = 2 * (size_t)n_centers * 4;
    TM_FORWARD(tmx::scene_pinned_ensure(s, head + 64));
    TM_HIP_CHECK(hipMemcpyAsync(s->h_pinned, s->ed_S.p, head, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    memcpy(counts, s->h_pinned, (size_t)n_centers * 4);
    memcpy(t_counts, (const uint32_t*)s->h_pinned + n_centers, (size_t)n_centers * 4);
    uint64_t total = 0;
    for (uint32_t k = 0; k < n_centers; ++k) total += t_counts[k];
    TM_REQUIRE(total <= 0xFFFFFFFFull, TM_ERR_UNSUPPORTED, "tm_scene_radius_tangent_lists: more than 2^32 pairs");
    *n_pairs = (uint32_t)total;
    if (total == 0) return TM_OK;
    TM_REQUIRE(pairs && cap_pairs >= total, TM_ERR_CAPACITY, "tm_scene_radius_tangent_lists: room for fewer pairs than found");
    const size_t need = (size_t)total * 8;
    TM_FORWARD(tmx::scene_pinned_ensure(s, need));
    TM_HIP_CHECK(hipMemcpyAsync(s->h_pinned, d_csr, need, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    memcpy(pairs, s->h_pinned, need);
    return TM_OK;
}

extern "C" int tm_scene_radius_lists(tm_scene* s, const uint32_t* centers, uint32_t n_centers, float radius,
                                     int32_t* lists, uint32_t stride, uint32_t* counts) {
    TM_GUARD(s);
    TM_REQUIRE(s && counts && (centers || n_centers == 0), TM_ERR_INVALID, "tm_scene_radius_lists: null argument");
    if (n_centers == 0) return TM_OK;
    for (uint32_t k = 0; k < n_centers; ++k)
        TM_REQUIRE(centers[k] < s->n, TM_ERR_INVALID, "tm_scene_radius_lists: centre index out of range");
    tm_ctx* c = s->ctx;
    TM_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t mwords = (s->n + 63) / 64;
    TM_FORWARD(s->ed_centres.ensure(n_centers));
    TM_FORWARD(s->ed_member.ensure((size_t)n_centers * mwords));
    TM_FORWARD(s->ed_wprefix.ensure((size_t)n_centers * mwords));
    TM_FORWARD(s->ed_S.ensure(n_centers));
    TM_FORWARD(s->ed_cpos.ensure(n_centers));
    TM_HIP_CHECK(hipMemcpyAsync(s->ed_centres.p, centers, (size_t)n_centers * 4, hipMemcpyHostToDevice, c->stream));
    const dim3 grid_n((s->n + 255) / 256, n_centers);
    hipLaunchKernelGGL(k_ball_members, dim3(grid_n.x), dim3(256), 0, c->stream, s->opos.p, s->n, s->ed_centres.p, n_centers,
                       radius * radius, mwords, s->ed_member.p, s->ed_cpos.p);
    hipLaunchKernelGGL(k_ball_prefix, dim3(n_centers), dim3(1024), 0, c->stream, s->ed_member.p, mwords,
                       s->ed_wprefix.p, s->ed_S.p);
    TM_HIP_CHECK(hipGetLastError());
    TM_HIP_CHECK(hipMemcpyAsync(counts, s->ed_S.p, (size_t)n_centers * 4, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (!lists) return TM_OK; /* counts only */
    uint32_t S_max = 0;
    for (uint32_t k = 0; k < n_centers; ++k) S_max = std::max(S_max, counts[k]);
    /* counts are valid on TM_ERR_CAPACITY: the caller may retry with a larger stride */
    TM_REQUIRE(stride >= S_max, TM_ERR_CAPACITY, "tm_scene_radius_lists: stride smaller than a neighbourhood");
    if (S_max == 0) return TM_OK;
    TM_FORWARD(s->ed_subset.ensure((size_t)n_centers * S_max));
    hipLaunchKernelGGL(k_ball_list, grid_n, dim3(256), 0, c->stream, s->ed_member.p, s->ed_wprefix.p, s->n, mwords,
                       S_max, s->ed_subset.p);
    TM_HIP_CHECK(hipGetLastError());
    /* one copy into pinned staging memory (a pageable destination moves at a fraction of the PCIe rate), then the
     * rows go to the caller's buffer */
    const size_t need = (size_t)n_centers * S_max * 4;
    TM_FORWARD(tmx::scene_pinned_ensure(s, need));
    TM_HIP_CHECK(hipMemcpyAsync(s->h_pinned, s->ed_subset.p, need, hipMemcpyDeviceToHost, c->stream));
    TM_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (uint32_t k = 0; k < n_centers; ++k)
        if (counts[k])
            memcpy(lists + (size_t)k * stride, (const int32_t*)s->h_pinned + (size_t)k * S_max, (size_t)counts[k] * 4);
    return TM_OK;
}

