/*
 * triplet_match/search.hpp — backend-agnostic host orchestration of the RANSAC search.
 *
 * Restates scene<Point>::impl::find_all_parallel / find_parallel / find_in_subset / icp_ of the
 * reference (include/impl/scene.hpp:39-98, 108-240, 242-367, 369-404). All heavy work goes through a
 * Backend (query + hypothesis generation, batched scoring, correspondence extraction); the product
 * backend is the HIP C ABI (scene.hpp), tests also run this code over the CPU oracle to check the host
 * logic without a GPU.
 *
 * Where the reference is non-deterministic or third-party-defined this file fixes a definition
 * (SURVEY.md Appendix A, Q10):
 *   - RNG: std::mt19937 seeded with 13 (the reference's `deterministic` branch, :123,245); draws are
 *     `rng() % k` in a partial Fisher-Yates shuffle (range-v3's view::sample / action::shuffle are
 *     unpinned); the shuffle of a neighbourhood is run lazily, as far as the inner loop consumes it;
 *   - workers: the reference splits the outer sample over hardware_concurrency()-1 threads and each
 *     returns its own first sufficient hypothesis; `workers` is an explicit parameter here (default 7);
 *   - radius_search_inclusive(diameter, p1): membership d2 <= r*r, ascending index order.
 */
#ifndef TRIPLET_MATCH_SEARCH_HPP_
#define TRIPLET_MATCH_SEARCH_HPP_

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <type_traits>
#include <utility>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "worker_pool.hpp"

namespace triplet_match {
namespace search {

constexpr double match_probability = 0.999;   /* include/impl/scene.hpp:14 */
constexpr uint64_t min_sample_count = 10ull;  /* :15 */
constexpr int query_limit = 200;              /* :19 */

struct pose_t {
    float m[16];
};

struct candidate_t { /* match_t with the scene->model transform (inverted once at :92) */
    pose_t transform;
    std::vector<uint32_t> scene_corrs, model_corrs;
    double signed_score = 0.0;
    int32_t device_row = -1; /* >= 0: the backend still holds these lists (row of its last batched extraction) */
};

/* wall-clock seconds per phase of a search, filled when params_t::stats points at one */
struct stats_t {
    double radius = 0, shuffle = 0, sample = 0, generate_score = 0, extract = 0, icp = 0, arbitration = 0,
           mask = 0, total = 0;
    /* how much sampling a search did: first points whose neighbourhood was sampled, positions drawn from the
     * shuffled neighbourhoods, neighbourhood entries the backend handed over, find_parallel rounds */
    uint64_t firsts_sampled = 0, draws = 0, neighbours = 0, rounds = 0;
};

/* What one chunk of first points (one worker of the reference, include/impl/scene.hpp:156-166) hands to
 * the arbitration: whether it found a sufficient hypothesis, how many first points it considered (they
 * are marked in fp_mask_, :160-162) and the candidate's pose after ICP. 96 bytes, the layout of the C
 * ABI's tm_best_record, so that tm_comm_allgather_records can carry it between ranks. */
struct chunk_record {
    uint32_t considered = 0;
    uint32_t found = 0; /* 1 = pose is a candidate */
    double reserved = 0.0;
    float pose[16] = {0};
    uint32_t pad[4] = {0, 0, 0, 0};
};
static_assert(sizeof(chunk_record) == 96, "chunk_record must match tm_best_record");

/* Exchange concept (multi-GPU search, one process per GPU, every rank holds a replica of scene and model):
 *   int rank(); int world();
 *   void allgather(const chunk_record* local, uint32_t n, chunk_record* all);  // all[r * n + k]
 * single_rank is the trivial one. */
struct single_rank {
    int rank() const { return 0; }
    int world() const { return 1; }
    void allgather(const chunk_record* local, uint32_t n, chunk_record* all) const {
        for (uint32_t k = 0; k < n; ++k) all[k] = local[k];
    }
};

/* What a search did, entry by entry, filled when params_t::trace points at one (tests compare it with the traces of
 * the reference text restated in Python, tests/golden/search_trace_*.json.gz): per find_parallel round the RANSAC
 * bound and the sampled first points (:131-150), per chunk and first point the neighbourhood size, inner bound
 * (:276-279), positions drawn and valid second points (:288-305), the chunk's first sufficient hypothesis (:330-338)
 * and `cons` (:271), the candidates after icp_ (:169-171) and the accepted matches (:200-237). */
struct trace_first {
    uint32_t i = 0, nn = 0;
    uint64_t inner_bound = 0, draws = 0;
    bool sampled = false; /* false: skipped because the chunk already had its candidate */
    std::vector<uint32_t> js;
};
struct trace_chunk {
    std::vector<uint32_t> firsts_in;
    std::vector<trace_first> firsts;
    uint32_t cons = 0;
    bool hit = false;
    pose_t hit_pose;
};
struct trace_match {
    pose_t transform;
    std::vector<uint32_t> scene_corrs, model_corrs;
    double signed_score = 0.0;
};
struct trace_round {
    uint64_t left = 0, outer_bound = 0, batch_size = 0;
    uint32_t min_points = 0, min_points_all = 0;
    std::vector<uint32_t> outer;
    std::vector<trace_chunk> chunks;
    std::vector<trace_match> candidates, accepted;
};
struct trace_t {
    std::vector<trace_round> rounds;
};

struct params_t {
    float dist_thres = 3.f;
    float accept_prob = 0.5f; /* model_match_factor */
    float early_out_factor = 0.9f;
    float min_diameter_factor = 0.2f, max_diameter_factor = 1.0f;
    bool force_up = false;
    uint32_t max_icp_iterations = 0;
    uint32_t workers = 7;
    uint64_t seed = 13;
    bool early_drop = true; /* detail::early_drop, :25 */
    /* Projected searches only (0 = the reference's rule): a hypothesis becomes THE candidate of its chunk
     * (:330-338) only if, besides #corrs > accept_prob * |model cloud|, at least this many of the scene's
     * tangent points are correspondences. On a projector's surface (a pipe, a wall) plain points match plain
     * points under any in-surface pose, so the count alone is reached by almost every hypothesis and the
     * first of them would be taken; the arbitration would then reject it for lack of tangent inliers
     * (min_points, :114,227) and the round would end empty. The driver sets this to that same min_points;
     * among the hypotheses of one scoring call that qualify, the one with most tangent inliers is taken. */
    uint32_t min_tangent_inliers = 0;
    bool verbose = false;   /* progress lines on stderr (the reference prints pdebug lines) */
    stats_t* stats = nullptr;
    trace_t* trace = nullptr;             /* whole search */
    trace_chunk* trace_chunk_ = nullptr;  /* set per chunk by find_parallel */
};

/*
 * Backend concept:
 *   uint32_t scene_size(); uint32_t model_size(); uint32_t model_point_count(); float model_diameter();
 *   const float* scene_surfel(uint32_t i);            // 12 floats
 *   bool pair_is_valid_sample(uint32_t i, uint32_t j);// filters :290-297 + feature/valid :299-302
 *   void announce_firsts(const std::vector<uint32_t>& firsts); // the points radius_subset will be asked for, in order
 *   void radius_subset(uint32_t i, float radius, std::vector<int32_t>& out);
 *   void generate(const std::vector<uint32_t>& ij, std::vector<pose_t>& poses);   // visit order
 *   void score(const std::vector<pose_t>& poses, const std::vector<int32_t>& subset, bool early_drop,
 *              std::vector<uint32_t>& counts, std::vector<double>& scores);       // project_ per pose
 *   bool supports_batched();                          // optional fast path:
 *   void generate_with_pairs(ij, poses, pair_of_hyp); //   hypotheses + the sample pair each came from
 *   void score_ball(poses, centers, radius, early_drop, counts, scores); // each pose on nn(its first point)
 *   void extract(const pose_t& T, float dist_thres, candidate_t& out);            // finish_find
 *   detail::icp_sums_t icp_sums(const pose_t& T, float dist_thres);               // finish_find + ICP sums
 *   void set_mask(const std::vector<int32_t>& mask);
 */

namespace detail {

/* adds the lifetime of the object to *acc (no-op for nullptr) */
struct timed {
    double* acc;
    std::chrono::steady_clock::time_point t0;
    explicit timed(double* a) : acc(a) {
        if (acc) t0 = std::chrono::steady_clock::now();
    }
    ~timed() {
        if (acc) *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
};
#define TM_TIMED(field) ::triplet_match::search::detail::timed _tm_timer_##field(p.stats ? &p.stats->field : nullptr)

/* ICP sums: count n and, over the correspondences (s = scene point, m = model point), sum s (3),
 * sum m (3), sum s_a*m_b (9, index 3*a+b), each term truncated to 2^-32 fixed point and added as
 * integers (order-free: a GPU reduction and a host loop give the same 15 numbers). */
struct icp_sums_t {
    uint32_t n = 0;
    long long v[15] = {0};
};
inline void icp_sums_add(icp_sums_t& acc, const float* s, const float* m) {
    const double S = 4294967296.0;
    for (int c = 0; c < 3; ++c) {
        acc.v[c] += (long long)((double)s[c] * S);
        acc.v[3 + c] += (long long)((double)m[c] * S);
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) acc.v[6 + 3 * r + c] += (long long)(((double)s[r] * (double)m[c]) * S);
    ++acc.n;
}

/* Horn's closed form (unit quaternion = dominant eigenvector of the 4x4 N matrix, Jacobi sweeps in
 * double): rigid T minimising sum |m - T*s|^2, the no-scale case of Eigen::umeyama (:393) */
inline pose_t rigid_fit(const icp_sums_t& in) {
    const double inv = 1.0 / 4294967296.0, n = (double)in.n;
    double cs[3], cm[3], S[3][3];
    for (int c = 0; c < 3; ++c) {
        cs[c] = (double)in.v[c] * inv / n;
        cm[c] = (double)in.v[3 + c] * inv / n;
    }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) S[a][b] = (double)in.v[6 + 3 * a + b] * inv - n * cs[a] * cm[b];
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) off += N[p][q] * N[p][q];
        if (off < 1e-30) break;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (std::fabs(N[p][q]) < 1e-300) continue;
                double theta = (N[q][q] - N[p][p]) / (2.0 * N[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {
                    double a = N[k][p], b = N[k][q];
                    N[k][p] = c * a - s * b;
                    N[k][q] = s * a + c * b;
                }
                for (int k = 0; k < 4; ++k) {
                    double a = N[p][k], b = N[q][k];
                    N[p][k] = c * a - s * b;
                    N[q][k] = s * a + c * b;
                }
                for (int k = 0; k < 4; ++k) {
                    double a = V[k][p], b = V[k][q];
                    V[k][p] = c * a - s * b;
                    V[k][q] = s * a + c * b;
                }
            }
    }
    int best = 0;
    for (int k = 1; k < 4; ++k)
        if (N[k][k] > N[best][best]) best = k;
    double q0 = V[0][best], qx = V[1][best], qy = V[2][best], qz = V[3][best];
    double nq = std::sqrt(q0 * q0 + qx * qx + qy * qy + qz * qz);
    q0 /= nq, qx /= nq, qy /= nq, qz /= nq;
    double R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * q0), 2 * (qx * qz + qy * q0)},
                      {2 * (qx * qy + qz * q0), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * q0)},
                      {2 * (qx * qz - qy * q0), 2 * (qy * qz + qx * q0), 1 - 2 * (qx * qx + qy * qy)}};
    pose_t T;
    for (int i = 0; i < 16; ++i) T.m[i] = 0.f;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T.m[c * 4 + r] = (float)R[r][c];
        T.m[12 + r] = (float)(cm[r] - (R[r][0] * cs[0] + R[r][1] * cs[1] + R[r][2] * cs[2]));
    }
    T.m[15] = 1.f;
    return T;
}

/* first `count` elements become a uniform sample in random order (sample + shuffle, :143-144) */
template <typename T>
inline void partial_shuffle(std::vector<T>& v, size_t count, std::mt19937& rng) {
    const size_t n = v.size();
    for (size_t k = 0; k < count && k + 1 < n; ++k) {
        size_t j = k + (size_t)(rng() % (uint32_t)(n - k));
        std::swap(v[k], v[j]);
    }
}

/* does the backend offer the whole batch step (generate + neighbourhood scoring + first sufficient
 * hypothesis) as one call? (hip_backend: tm_search_batch) */
template <class B, class = void>
struct has_search_batch : std::false_type {};
template <class B>
struct has_search_batch<B, decltype(void(std::declval<B&>().search_batch(
                               std::declval<const std::vector<uint32_t>&>(), 0.f, false, 0.f, std::declval<pose_t&>())))>
    : std::true_type {};
template <class B>
inline bool fused_batch(B& be, const std::vector<uint32_t>& ij, float radius, bool early_drop, float sufficient,
                        pose_t& pose, bool& found, std::true_type) {
    if (!be.supports_fused_batch()) return false;
    found = be.search_batch(ij, radius, early_drop, sufficient, pose);
    return true;
}
template <class B>
inline bool fused_batch(B&, const std::vector<uint32_t>&, float, bool, float, pose_t&, bool&, std::false_type) {
    return false;
}

/* can the backend take the batches of a whole round in one call? (hip_backend: tm_search_batches) */
template <class B, class = void>
struct has_search_batches : std::false_type {};
template <class B>
struct has_search_batches<B, decltype(void(std::declval<B&>().search_batches(
                                 std::declval<const std::vector<std::vector<uint32_t>>&>(), 0.f, false, 0.f,
                                 std::declval<std::vector<char>&>(), std::declval<std::vector<pose_t>&>())))>
    : std::true_type {};
template <class B>
inline bool deferred_batches(B& be, std::true_type) { return be.supports_deferred_batches(); }
template <class B>
inline bool deferred_batches(B&, std::false_type) { return false; }
template <class B>
inline void run_batches(B& be, const std::vector<std::vector<uint32_t>>& batches, float radius, bool early_drop,
                        float sufficient, std::vector<char>& found, std::vector<pose_t>& poses, std::true_type) {
    be.search_batches(batches, radius, early_drop, sufficient, found, poses);
}
template <class B>
inline void run_batches(B&, const std::vector<std::vector<uint32_t>>&, float, bool, float, std::vector<char>&,
                        std::vector<pose_t>&, std::false_type) {}

/* does the backend hand out its neighbourhood lists in place? (hip_backend: a view into the lists it fetched
 * for 16 first points at once; saves a 0.4 MB copy per first point at the C2 size) */
template <class B, class = void>
struct has_radius_view : std::false_type {};
template <class B>
struct has_radius_view<B, decltype(void(std::declval<B&>().radius_view(0u, 0.f, std::declval<const int32_t*&>(),
                                                                       std::declval<size_t&>())))> : std::true_type {};
template <class B>
inline void neighbourhood(B& be, uint32_t i, float radius, std::vector<int32_t>&, const int32_t*& ptr, size_t& n,
                          std::true_type) {
    be.radius_view(i, radius, ptr, n);
}
template <class B>
inline void neighbourhood(B& be, uint32_t i, float radius, std::vector<int32_t>& store, const int32_t*& ptr, size_t& n,
                          std::false_type) {
    be.radius_subset(i, radius, store);
    ptr = store.data();
    n = store.size();
}

/* does the backend hand out the sampling loop's view of a neighbourhood — its size and its tangent members with
 * their ranks — instead of the whole list? (hip_backend: tm_scene_radius_tangent_lists) */
template <class B, class = void>
struct has_tangent_view : std::false_type {};
template <class B>
struct has_tangent_view<B, decltype(void(std::declval<B&>().tangent_view(0u, 0.f, std::declval<size_t&>(),
                                                                         std::declval<const uint32_t*&>(),
                                                                         std::declval<size_t&>())))> : std::true_type {};
template <class B>
inline bool tangent_neighbourhood(B& be, uint32_t i, float radius, size_t& n_nn, const uint32_t*& pairs, size_t& n_pairs,
                                  std::true_type) {
    if (!be.supports_tangent_view()) return false;
    be.tangent_view(i, radius, n_nn, pairs, n_pairs);
    return true;
}
template <class B>
inline bool tangent_neighbourhood(B&, uint32_t, float, size_t&, const uint32_t*&, size_t&, std::false_type) {
    return false;
}
template <class B, class = void>
struct has_announce_round : std::false_type {};
template <class B>
struct has_announce_round<B, decltype(void(std::declval<B&>().announce_round(std::declval<const std::vector<uint32_t>&>())))>
    : std::true_type {};
template <class B>
inline void announce_round(B& be, const std::vector<uint32_t>& firsts, std::true_type) { be.announce_round(firsts); }
template <class B>
inline void announce_round(B&, const std::vector<uint32_t>&, std::false_type) {}

/* can the chunks of an announced round be sampled on concurrent threads? A backend says so when the neighbourhood
 * views of every announced first point are in its hands (round_views_ready(radius) fetches them and reports) and
 * reading them, pair_is_valid_sample and the surfel accessors neither modify it nor take a lock */
template <class B, class = void>
struct has_round_views_ready : std::false_type {};
template <class B>
struct has_round_views_ready<B, decltype(void(std::declval<B&>().round_views_ready(0.f)))> : std::true_type {};
template <class B>
inline bool round_views_ready(B& be, float radius, std::true_type) { return be.round_views_ready(radius); }
template <class B>
inline bool round_views_ready(B&, float, std::false_type) { return false; }

/* does the backend run finish_find for several poses in one call? (hip_backend: tm_extract_correspondences_batch) */
template <class B, class = void>
struct has_extract_batch : std::false_type {};
template <class B>
struct has_extract_batch<B, decltype(void(std::declval<B&>().extract_batch(std::declval<const std::vector<pose_t>&>(), 0.f,
                                                                             std::declval<std::vector<candidate_t>&>())))>
    : std::true_type {};
template <class B>
inline void extract_all(B& be, const std::vector<pose_t>& Ts, float dist_thres, std::vector<candidate_t>& out, std::true_type) {
    be.extract_batch(Ts, dist_thres, out);
}
template <class B>
inline void extract_all(B& be, const std::vector<pose_t>& Ts, float dist_thres, std::vector<candidate_t>& out, std::false_type) {
    out.resize(Ts.size());
    for (size_t k = 0; k < Ts.size(); ++k) be.extract(Ts[k], dist_thres, out[k]);
}

/* The shuffled copy of a neighbourhood list, kept sparse. Of the list only the entries with tangent_mask set matter
 * (every other one fails the first filter of include/impl/scene.hpp:290 whatever its index is); they arrive as
 * (rank, index) pairs in ascending rank order. Before any swap, position p holds pairs[ordinal(p)] if bit p of
 * `tangent` is set (ordinal = number of set bits below p: a per-word prefix count + a popcount), else "nothing" (-1).
 * A swap of the lazy Fisher-Yates shuffle that involves a tangent entry records the two positions' new contents in a
 * small open-addressing table (`moved` marks them); swapping two positions that hold nothing is nothing. Per first
 * point: clearing two bitmaps of |nn| bits, setting the tangent bits, one prefix pass — no per-entry inserts. */
struct sparse_list {
    const uint32_t* pairs = nullptr; /* (rank, index), ascending rank */
    std::vector<uint64_t> tangent, moved;
    std::vector<uint32_t> word_prefix; /* tangent bits set before each word */
    std::vector<uint32_t> key;         /* displaced positions */
    std::vector<int32_t> val;
    uint32_t shift = 32, used = 0;
    static constexpr uint32_t kEmpty = 0xFFFFFFFFu;
    void reset(const uint32_t* rank_index_pairs, size_t n_pairs, size_t positions) {
        pairs = rank_index_pairs;
        const size_t words = (positions + 63) / 64 + 1;
        tangent.assign(words, 0ull);
        moved.assign(words, 0ull);
        for (size_t e = 0; e < n_pairs; ++e) tangent[pairs[2 * e] >> 6] |= 1ull << (pairs[2 * e] & 63u);
        word_prefix.resize(words);
        uint32_t run = 0;
        for (size_t w = 0; w < words; ++w) {
            word_prefix[w] = run;
            run += (uint32_t)__builtin_popcountll(tangent[w]);
        }
        if (key.size() != 2048) {
            key.assign(2048, kEmpty);
            val.resize(2048);
            shift = 32 - 11;
        } else if (used) {
            std::fill(key.begin(), key.end(), kEmpty);
        }
        used = 0;
    }
    uint32_t slot_of(uint32_t pos) const { return (pos * 0x9E3779B1u) >> shift; }
    int32_t get(uint32_t pos) const {
        const uint64_t bit = 1ull << (pos & 63u);
        if (!(moved[pos >> 6] & bit)) { /* untouched by the shuffle so far */
            const uint64_t tw = tangent[pos >> 6];
            if (!(tw & bit)) return -1;
            return (int32_t)pairs[2 * (size_t)(word_prefix[pos >> 6] + (uint32_t)__builtin_popcountll(tw & (bit - 1))) + 1];
        }
        const uint32_t mask = (uint32_t)key.size() - 1;
        for (uint32_t h = slot_of(pos);; h = (h + 1) & mask) {
            if (key[h] == pos) return val[h];
            if (key[h] == kEmpty) return -1;
        }
    }
    void put(uint32_t pos, int32_t v) {
        moved[pos >> 6] |= 1ull << (pos & 63u);
        const uint32_t mask = (uint32_t)key.size() - 1;
        for (uint32_t h = slot_of(pos);; h = (h + 1) & mask) {
            if (key[h] == pos) {
                val[h] = v;
                return;
            }
            if (key[h] == kEmpty) {
                key[h] = pos;
                val[h] = v;
                if (++used * 2 > key.size()) grow();
                return;
            }
        }
    }
    void grow() {
        std::vector<uint32_t> k2;
        std::vector<int32_t> v2;
        k2.swap(key);
        v2.swap(val);
        key.assign(k2.size() * 2, kEmpty);
        val.resize(k2.size() * 2);
        --shift;
        used = 0;
        const uint32_t mask = (uint32_t)key.size() - 1;
        for (size_t q = 0; q < k2.size(); ++q) {
            if (k2[q] == kEmpty) continue;
            uint32_t h = slot_of(k2[q]);
            while (key[h] != kEmpty) h = (h + 1) & mask;
            key[h] = k2[q];
            val[h] = v2[q];
            ++used;
        }
    }
};

/* does the backend take the two changes find_all makes to mask_ — all clear (:41), an accepted match's points set
 * (:87-90) — as such, instead of the whole mask every time? (hip_backend: a device memset, tm_scene_mask_add) */
template <class B, class = void>
struct has_sparse_mask : std::false_type {};
template <class B>
struct has_sparse_mask<B, decltype(void(std::declval<B&>().clear_mask()),
                                   void(std::declval<B&>().add_to_mask(std::declval<const std::vector<uint32_t>&>())))>
    : std::true_type {};
template <class B>
inline void mask_clear(B& be, const std::vector<int32_t>&, std::true_type) { be.clear_mask(); }
template <class B>
inline void mask_clear(B& be, const std::vector<int32_t>& mask, std::false_type) { be.set_mask(mask); }
template <class B>
inline void mask_add(B& be, const std::vector<int32_t>&, const std::vector<uint32_t>& added, std::true_type) { be.add_to_mask(added); }
template <class B>
inline void mask_add(B& be, const std::vector<int32_t>& mask, const std::vector<uint32_t>&, std::false_type) { be.set_mask(mask); }

/* does the backend extract AND arbitrate the candidates of a round where their lists are (hip_backend:
 * tm_extract_arbitrate_batch)? true = `accepted` is what arbitrate() would return for these poses, in its order, with
 * device_row set; false = not decided there (unsupported, switched off, or an accepted candidate needs trimming) */
template <class B, class = void>
struct has_extract_arbitrate : std::false_type {};
template <class B>
struct has_extract_arbitrate<B, decltype(void(std::declval<B&>().extract_arbitrate(std::declval<const std::vector<pose_t>&>(), 0.f, 0u, 0u,
                                                                                    std::declval<std::vector<candidate_t>&>(), false)))>
    : std::true_type {};
template <class B>
inline bool extract_arbitrate(B& be, const std::vector<pose_t>& Ts, float dist_thres, uint32_t min_points, uint32_t min_points_all,
                              std::vector<candidate_t>& accepted, bool verbose, std::true_type) {
    return be.extract_arbitrate(Ts, dist_thres, min_points, min_points_all, accepted, verbose);
}
template <class B>
inline bool extract_arbitrate(B&, const std::vector<pose_t>&, float, uint32_t, uint32_t, std::vector<candidate_t>&, bool, std::false_type) {
    return false;
}
/* ... and set mask_ at the points of rows of that extraction? */
template <class B, class = void>
struct has_mask_add_extracted : std::false_type {};
template <class B>
struct has_mask_add_extracted<B, decltype(void(std::declval<B&>().add_extracted_to_mask(std::declval<const std::vector<uint32_t>&>())))>
    : std::true_type {};
template <class B>
inline void mask_add_extracted(B& be, const std::vector<uint32_t>& rows, std::true_type) { be.add_extracted_to_mask(rows); }
template <class B>
inline void mask_add_extracted(B&, const std::vector<uint32_t>&, std::false_type) {}

/* does the backend run the whole ICP loop itself? (hip_backend: tm_icp_refine) */
template <class B, class = void>
struct has_icp_refine : std::false_type {};
template <class B>
struct has_icp_refine<B, decltype(void(std::declval<B&>().icp_refine(std::declval<const pose_t&>(), 0.f, 0u)))>
    : std::true_type {};
template <class B>
inline bool device_icp(B& be, const pose_t& start, float dist_thres, uint32_t max_iterations, pose_t& out,
                       std::true_type) {
    if (!be.supports_icp_refine()) return false;
    out = be.icp_refine(start, dist_thres, max_iterations);
    return true;
}
template <class B>
inline bool device_icp(B&, const pose_t&, float, uint32_t, pose_t&, std::false_type) {
    return false;
}
/* does the backend refine several independent candidates in one call? (hip_backend: tm_icp_refine_batch) */
template <class B, class = void>
struct has_icp_refine_batch : std::false_type {};
template <class B>
struct has_icp_refine_batch<B, decltype(void(std::declval<B&>().icp_refine_batch(std::declval<const std::vector<pose_t>&>(), 0.f, 0u,
                                                                                   std::declval<std::vector<pose_t>&>())))>
    : std::true_type {};
template <class B>
inline bool device_icp_batch(B& be, const std::vector<pose_t>& starts, float dist_thres, uint32_t max_iterations,
                             std::vector<pose_t>& out, std::true_type) {
    if (!be.supports_icp_refine_batch()) return false;
    be.icp_refine_batch(starts, dist_thres, max_iterations, out);
    return true;
}
template <class B>
inline bool device_icp_batch(B&, const std::vector<pose_t>&, float, uint32_t, std::vector<pose_t>&, std::false_type) {
    return false;
}

}  // namespace detail

/* find_in_subset, include/impl/scene.hpp:242-367 */
template <class Backend>
inline std::pair<std::vector<candidate_t>, uint32_t> find_in_subset(
    Backend& be, const std::vector<uint32_t>& firsts, const std::vector<int32_t>& mask,
    const std::vector<int32_t>& fp_mask, const std::vector<int32_t>& tangent_mask, const params_t& p,
    std::vector<std::vector<uint32_t>>* recorder = nullptr) {
    /* recorder: the batches are not scored here but handed back in visit order (find_parallel runs the batches of
     * all its chunks in one call; the first batch of a chunk with a sufficient hypothesis is the chunk's result —
     * the same as stopping at it, because what a batch contains does not depend on earlier batches' scores) */
    std::mt19937 rng;
    rng.seed((uint32_t)p.seed);
    std::vector<pose_t> best_ts;
    uint32_t cons = 0;
    const uint32_t n_model_all = be.model_size();
    const float sufficient = p.accept_prob * (float)n_model_all; /* :331, size_t > float in float */
    std::vector<int32_t> nn, inner0;
    detail::sparse_list shuffled;
    const int32_t* nn_ptr = nullptr; /* the neighbourhood of the current first point (in nn, or the backend's) */
    size_t n_nn = 0;
    std::vector<uint32_t> ij;
    std::vector<pose_t> poses;
    std::vector<uint32_t> counts;
    std::vector<double> scores;
    /* Batched path: the valid samples of up to 16 first points go through one generate + one scoring
     * call in which every hypothesis is scored on the neighbourhood of its own first point (with the
     * early-drop rule applied per hypothesis on that neighbourhood's scan order when p.early_drop);
     * scanning the results in visit order gives the same first sufficient hypothesis as the
     * one-first-point-at-a-time loop below. */
    /* first hypothesis in visit order that is sufficient (:330-338); projected searches: the sufficient one
     * with most tangent inliers, at least min_tangent_inliers */
    std::vector<int32_t> tangent_subset; /* filled when the first candidate needs it: a pass over the whole scene */
    bool have_tangent_subset = false;
    auto first_sufficient = [&](const std::vector<pose_t>& ps, const std::vector<uint32_t>& cs,
                                const std::vector<double>& ss) -> long {
        std::vector<uint32_t> cand;
        for (size_t h = 0; h < ps.size(); ++h)
            if (ss[h] > 0.0 && (float)cs[h] > sufficient) {
                if (!p.min_tangent_inliers) return (long)h;
                cand.push_back((uint32_t)h);
            }
        if (cand.empty()) return -1;
        if (!have_tangent_subset) {
            for (uint32_t i = 0; i < (uint32_t)tangent_mask.size(); ++i)
                if (tangent_mask[i] && !mask[i]) tangent_subset.push_back((int32_t)i);
            have_tangent_subset = true;
        }
        if (tangent_subset.empty()) return -1;
        std::vector<pose_t> cp(cand.size());
        for (size_t k = 0; k < cand.size(); ++k) cp[k] = ps[cand[k]];
        std::vector<uint32_t> tc;
        std::vector<double> ts;
        be.score(cp, tangent_subset, false, tc, ts); /* correspondences among the tangent points alone */
        long best = -1; /* the sufficient hypothesis with most tangent inliers (first of equals) */
        uint32_t best_tc = 0;
        for (size_t k = 0; k < cand.size(); ++k)
            if (tc[k] >= p.min_tangent_inliers && (best < 0 || tc[k] > best_tc)) best = (long)cand[k], best_tc = tc[k];
        return best;
    };
    const bool batched = be.supports_batched();
    std::vector<uint32_t> batch_ij, pair_of_hyp, centers;
    auto flush_batch = [&]() {
        if (batch_ij.empty() || !best_ts.empty()) {
            batch_ij.clear();
            return;
        }
        if (recorder) {
            recorder->push_back(batch_ij);
            batch_ij.clear();
            return;
        }
        TM_TIMED(generate_score);
        if (!p.min_tangent_inliers) { /* the whole step on the device when the backend has it */
            pose_t hit;
            bool found = false;
            if (detail::fused_batch(be, batch_ij, be.model_diameter(), p.early_drop, sufficient, hit, found,
                                    detail::has_search_batch<Backend>{})) {
                if (found) {
                    best_ts.push_back(hit);
                    if (p.trace_chunk_) p.trace_chunk_->hit = true, p.trace_chunk_->hit_pose = hit;
                }
                batch_ij.clear();
                return;
            }
        }
        be.generate_with_pairs(batch_ij, poses, pair_of_hyp);
        if (!poses.empty()) {
            centers.resize(poses.size());
            for (size_t h = 0; h < poses.size(); ++h) centers[h] = batch_ij[2 * (size_t)pair_of_hyp[h]];
            be.score_ball(poses, centers, be.model_diameter(), p.early_drop, counts, scores);
            const long h = first_sufficient(poses, counts, scores);
            if (h >= 0) {
                best_ts.push_back(poses[(size_t)h]);
                if (p.trace_chunk_) p.trace_chunk_->hit = true, p.trace_chunk_->hit_pose = poses[(size_t)h];
            }
        }
        batch_ij.clear();
    };
    uint32_t firsts_in_batch = 0;
    {
        std::vector<uint32_t> live; /* the first points the loop will actually look at (:268) */
        for (uint32_t i : firsts)
            if (!fp_mask[i] && !mask[i]) live.push_back(i);
        be.announce_firsts(live);
    }
    for (uint32_t i : firsts) {
        if (fp_mask[i] || mask[i]) continue; /* :268 */
        ++cons;
        const uint32_t* t_pairs = nullptr; /* sparse view: (rank, index) of the tangent members of nn */
        size_t n_t_pairs = 0;
        bool sparse = false;
        {
            TM_TIMED(radius);
            sparse = detail::tangent_neighbourhood(be, i, be.model_diameter(), n_nn, t_pairs, n_t_pairs,
                                                   detail::has_tangent_view<Backend>{});
            if (!sparse)
                detail::neighbourhood(be, i, be.model_diameter(), nn, nn_ptr, n_nn, detail::has_radius_view<Backend>{}); /* :273 */
        }
        if (n_nn == 0) continue;
        double prob = (double)n_model_all / (double)n_nn;
        uint64_t inner_bound = (uint64_t)(-std::log(1.0 - match_probability) / prob); /* :277 */
        inner_bound = std::max(inner_bound, min_sample_count);
        inner_bound = std::min<uint64_t>(inner_bound, n_nn);
        trace_first* tf = nullptr;
        if (p.trace_chunk_) {
            p.trace_chunk_->firsts.emplace_back();
            tf = &p.trace_chunk_->firsts.back();
            tf->i = i, tf->nn = (uint32_t)n_nn, tf->inner_bound = inner_bound;
        }
        if (!best_ts.empty()) continue; /* the reference keeps looping without effect once a candidate
                                           exists (best_ts.empty() gates :331; early-out is unreachable,
                                           SURVEY Q8): skip the dead work, keep `cons` */
        /* The shuffled neighbourhood (:285) is consumed from the front until inner_bound + 1 valid samples
         * are found (:305, :350), so the Fisher-Yates shuffle is run lazily: position k is drawn when the loop
         * reaches it. Same distribution as shuffling everything first; the number of draws per first point is
         * what this definition fixes (range-v3's action::shuffle is unpinned). */
        ij.clear();
        uint32_t valid_samples = 0;
        if (tf) tf->sampled = true;
        {
        TM_TIMED(sample);
        if (p.stats) {
            ++p.stats->firsts_sampled;
            p.stats->neighbours += sparse ? n_t_pairs : n_nn;
        }
        if (sparse) {
            /* the same shuffle on the sparse copy: the draws depend on the list's LENGTH only (rng() % (n - k)), and
             * a position's content matters only if it is a tangent point; same r, same j at every step as below */
            shuffled.reset(t_pairs, n_t_pairs, n_nn);
            for (size_t k = 0; k < n_nn; ++k) {
                if (p.stats) ++p.stats->draws;
                if (tf) ++tf->draws;
                if (k + 1 < n_nn) {
                    const size_t r = k + (size_t)(rng() % (uint32_t)(n_nn - k));
                    if (r != k) {
                        const int32_t vk = shuffled.get((uint32_t)k), vr = shuffled.get((uint32_t)r);
                        if (vk >= 0 || vr >= 0) {
                            shuffled.put((uint32_t)k, vr);
                            shuffled.put((uint32_t)r, vk);
                        }
                    }
                }
                const int32_t j = shuffled.get((uint32_t)k);
                if (j < 0) continue;                                             /* :290, tangent_mask_[j] is false */
                if (!tangent_mask[j] || mask[j] || i == (uint32_t)j) continue; /* :290 */
                if (!be.pair_is_valid_sample(i, (uint32_t)j)) continue;         /* :293-302 */
                ij.push_back(i);
                ij.push_back((uint32_t)j);
                ++valid_samples;
                if (valid_samples > inner_bound) break; /* :350 */
            }
        } else {
        inner0.assign(nn_ptr, nn_ptr + n_nn); /* a copy is shuffled (:284-285): project_ scans nn in its own order */
        for (size_t k = 0; k < n_nn; ++k) {
            if (p.stats) ++p.stats->draws;
            if (tf) ++tf->draws;
            if (k + 1 < n_nn) {
                const size_t r = k + (size_t)(rng() % (uint32_t)(n_nn - k));
                std::swap(inner0[k], inner0[r]);
            }
            const int32_t j = inner0[k];
            if (!tangent_mask[j] || mask[j] || i == (uint32_t)j) continue; /* :290 */
            if (!be.pair_is_valid_sample(i, (uint32_t)j)) continue;         /* :293-302 */
            ij.push_back(i);
            ij.push_back((uint32_t)j);
            ++valid_samples;
            if (valid_samples > inner_bound) break; /* :350 */
        }
        }
        }
        if (tf)
            for (size_t e = 1; e < ij.size(); e += 2) tf->js.push_back(ij[e]);
        if (ij.empty()) continue;
        if (batched) {
            batch_ij.insert(batch_ij.end(), ij.begin(), ij.end());
            if (++firsts_in_batch == 16) {
                flush_batch();
                firsts_in_batch = 0;
            }
            continue;
        }
        TM_TIMED(generate_score);
        be.generate(ij, poses); /* :304-319, visit order */
        if (poses.empty()) continue;
        if (sparse) /* one-first-point-at-a-time scoring scans the whole neighbourhood: fetch it now */
            be.radius_subset(i, be.model_diameter(), nn);
        else if (nn_ptr != nn.data() || nn.size() != n_nn)
            nn.assign(nn_ptr, nn_ptr + n_nn);
        be.score(poses, nn, p.early_drop, counts, scores); /* project_ on nn, :326 */
        /* :330-338: score > best_score(=0) and #corrs > accept_prob * |model cloud| */
        const long h = first_sufficient(poses, counts, scores);
        if (h >= 0) {
            best_ts.push_back(poses[(size_t)h]);
            if (p.trace_chunk_) p.trace_chunk_->hit = true, p.trace_chunk_->hit_pose = poses[(size_t)h];
        }
    }
    if (batched) flush_batch();
    if (p.trace_chunk_) p.trace_chunk_->cons = cons;
    std::vector<candidate_t> matches;
    TM_TIMED(extract);
    for (size_t k = 0; k < best_ts.size() && matches.size() < 5; ++k) { /* :361-364 */
        candidate_t c;
        be.extract(best_ts[k], p.dist_thres, c);
        matches.push_back(std::move(c));
    }
    return {matches, cons};
}

/* icp_, include/impl/scene.hpp:369-404 (note the doubled distance threshold, :373,395). Each pass needs
 * only the number of correspondences and their sums (be.icp_sums: finish_find + accumulation in one
 * go); the correspondence lists are extracted once, for the pose that is returned. */
template <class Backend>
inline candidate_t icp(Backend& be, const candidate_t& match, const params_t& p) {
    if (p.max_icp_iterations == 0) return match;
    TM_TIMED(icp);
    const float dt2 = 2.f * p.dist_thres;
    pose_t best_t = match.transform;
    {
        pose_t refined;
        if (detail::device_icp(be, best_t, dt2, p.max_icp_iterations, refined, detail::has_icp_refine<Backend>{})) {
            candidate_t out;
            be.extract(refined, dt2, out);
            return out;
        }
    }
    detail::icp_sums_t best = be.icp_sums(best_t, dt2);
    uint32_t iter = 0;
    for (;;) {
        if (best.n < 3) break; /* :381 */
        pose_t trans = detail::rigid_fit(best); /* :385-393 */
        detail::icp_sums_t proj = be.icp_sums(trans, dt2);
        if (proj.n < best.n) break; /* :396 */
        best = proj;
        best_t = trans;
        if (++iter == p.max_icp_iterations) break;
    }
    candidate_t out;
    be.extract(best_t, dt2, out);
    return out;
}

/* The arbitration of find_parallel, include/impl/scene.hpp:193-237: candidates sorted by their number of
 * correspondences (descending, stable), then accepted greedily: a candidate keeps only the scene points no
 * accepted candidate holds yet, and is accepted when at least min_points of those are tangent points and
 * at least min_points_all in total; a rejected candidate gives its points back (used_copy, :203).
 *
 * One pass over a candidate's list decides it: `owner` holds, per scene point, the number of the last candidate that
 * claimed it (0 = nobody); a point is free when its owner is nobody or a REJECTED candidate (a 1-entry-per-candidate
 * table says which), and a free point is claimed on the spot — a rejected candidate's claims simply stop counting,
 * nothing is given back by hand. An accepted candidate that lost points to earlier ones is compacted in a second pass
 * (its own number marks what it kept). Mark = the owner's type: a byte for up to 254 candidates. */
namespace detail {
template <class Mark>
inline std::vector<candidate_t> arbitrate_marks(std::vector<candidate_t>& cands, uint32_t n_scene_points,
                                                const std::vector<int32_t>& tangent_mask, uint32_t min_points,
                                                uint32_t min_points_all, bool verbose) {
    std::vector<candidate_t> result;
    static thread_local std::vector<Mark> owner; /* kept per thread: a fresh array per round is page faults */
    owner.assign(n_scene_points, Mark(0));
    std::vector<unsigned char> accepted(cands.size() + 1, 0);
    Mark* const own = owner.data();
    for (size_t ci = 0; ci < cands.size(); ++ci) { /* greedy disjoint accept, :200-237 */
        candidate_t& cand = cands[ci];
        const Mark me = (Mark)(ci + 1);
        const size_t n_c = cand.scene_corrs.size();
        const uint32_t* sc = cand.scene_corrs.data();
        uint32_t score = 0, kept = 0;
        for (size_t k = 0; k < n_c; ++k) {
            const uint32_t idx = sc[k];
            if (accepted[own[idx]]) continue; /* an accepted candidate holds it */
            own[idx] = me;
            ++kept;
            score += tangent_mask[idx] ? 1u : 0u;
        }
        const uint32_t score_all = kept;
        if (verbose)
            fprintf(stderr, "[search] candidate: %zu corrs (%u after overlap removal), tangent inliers %u (need %u), all %u (need %u)\n",
                    n_c, score_all, score, min_points, score_all, min_points_all);
        if (score < min_points || score_all < min_points_all) continue; /* :227: its claims stop counting */
        accepted[me] = 1;
        candidate_t out;
        out.transform = cand.transform;
        out.signed_score = cand.signed_score;
        if (kept == n_c) { /* overlaps nothing accepted: the candidate keeps its lists as they are */
            out.scene_corrs = std::move(cand.scene_corrs);
            out.model_corrs = std::move(cand.model_corrs);
        } else {
            const uint32_t* mc = cand.model_corrs.data();
            out.scene_corrs.resize(kept);
            out.model_corrs.resize(kept);
            uint32_t at = 0;
            for (size_t k = 0; k < n_c; ++k) {
                const uint32_t idx = sc[k];
                if (own[idx] != me) continue;
                out.scene_corrs[at] = idx;
                out.model_corrs[at] = mc[k];
                ++at;
            }
        }
        result.push_back(std::move(out));
    }
    return result;
}
}  // namespace detail

inline std::vector<candidate_t> arbitrate(std::vector<candidate_t> cands, uint32_t n_scene_points,
                                          const std::vector<int32_t>& tangent_mask, uint32_t min_points,
                                          uint32_t min_points_all, bool verbose = false) {
    std::stable_sort(cands.begin(), cands.end(), [](const candidate_t& a, const candidate_t& b) {
        return a.scene_corrs.size() > b.scene_corrs.size(); /* :193-195 */
    });
    if (cands.size() <= 254)
        return detail::arbitrate_marks<unsigned char>(cands, n_scene_points, tangent_mask, min_points, min_points_all, verbose);
    return detail::arbitrate_marks<uint32_t>(cands, n_scene_points, tangent_mask, min_points, min_points_all, verbose);
}

/* find_parallel, include/impl/scene.hpp:108-240 */
template <class Backend, class Exchange>
inline std::vector<candidate_t> find_parallel(Backend& be, const std::vector<uint32_t>& tangent_indices,
                                              uint32_t n_scene, std::vector<int32_t>& mask,
                                              std::vector<int32_t>& fp_mask,
                                              const std::vector<int32_t>& tangent_mask, const params_t& p,
                                              Exchange& ex) {
    const uint32_t n_model = be.model_point_count();
    const uint32_t n_model_all = be.model_size();
    const uint32_t min_points = (uint32_t)((1.f * p.accept_prob) * (float)n_model);      /* :114 */
    const uint32_t min_points_all = (uint32_t)(p.accept_prob * (float)n_model_all);      /* :115 */
    std::mt19937 rng;
    rng.seed((uint32_t)p.seed);
    std::vector<uint32_t> left;
    for (uint32_t idx : tangent_indices)
        if (!fp_mask[idx] && !mask[idx]) left.push_back(idx); /* :131 */
    trace_round* tr = nullptr;
    if (p.trace) {
        p.trace->rounds.emplace_back();
        tr = &p.trace->rounds.back();
        tr->left = left.size(), tr->min_points = min_points, tr->min_points_all = min_points_all;
    }
    if (left.empty()) return {};
    double outer_prob = (double)n_model / (double)left.size();
    uint64_t outer_bound;
    if (outer_prob >= 1.0)
        outer_bound = 0; /* log(1 - p) is undefined; the clamp below takes over */
    else
        outer_bound = (uint64_t)(std::log(1.0 - match_probability) / std::log(1.0 - outer_prob)); /* :134 */
    outer_bound = std::max(outer_bound, min_sample_count);
    outer_bound = std::min<uint64_t>(outer_bound, n_scene);
    outer_bound = std::min<uint64_t>(outer_bound, left.size());
    if (p.verbose)
        fprintf(stderr, "[search] outer bound: %llu of %zu unmasked tangent points\n", (unsigned long long)outer_bound,
                left.size()); /* the reference's pdebug("outer bound"), :152 */
    detail::partial_shuffle(left, (size_t)outer_bound, rng); /* sample + shuffle, :143-144 */
    std::vector<uint32_t> outer(left.begin(), left.begin() + (size_t)outer_bound);
    const uint32_t concurrency = std::max(1u, p.workers); /* :146 */
    uint64_t batch = outer_bound / concurrency;
    if (outer_bound % concurrency > 0) ++batch;
    if (tr) {
        tr->outer_bound = outer_bound, tr->batch_size = batch, tr->outer = outer;
        tr->chunks.resize(batch ? (size_t)((outer.size() + batch - 1) / batch) : 0);
        for (size_t ci = 0; ci < tr->chunks.size(); ++ci)
            tr->chunks[ci].firsts_in.assign(outer.begin() + ci * batch,
                                            outer.begin() + std::min<uint64_t>((ci + 1) * batch, outer.size()));
    }
    /* the parameters of chunk ci: the caller's, with that chunk's trace slot */
    auto chunk_params = [&](size_t ci) {
        params_t pl = p;
        pl.trace_chunk_ = tr ? &tr->chunks[ci] : nullptr;
        return pl;
    };
    std::vector<candidate_t> cands;
    if (ex.world() > 1) {
        /* Several ranks: the chunks are independent of one another inside a round (a chunk reads fp_mask_
         * only at its own first points, :268, and nothing else changes before the arbitration), so chunk c
         * goes to rank c mod world. Every rank searches and refines its chunks, ONE all-gather of 96-byte
         * chunk records follows, and every rank rebuilds all candidates (finish_find of the exchanged pose)
         * in chunk order: the arbitration below then sees the same list on every rank, and the same list as
         * a single rank would, whatever the number of ranks. */
        const uint32_t world = (uint32_t)ex.world(), rank = (uint32_t)ex.rank();
        const uint32_t n_chunks = (uint32_t)((outer.size() + batch - 1) / batch);
        const uint32_t per_rank = (n_chunks + world - 1) / world;
        std::vector<chunk_record> local(per_rank), all((size_t)per_rank * world);
        std::vector<std::vector<uint32_t>> chunks(n_chunks);
        for (uint32_t ci = 0; ci < n_chunks; ++ci) {
            const uint64_t c0 = (uint64_t)ci * batch;
            chunks[ci].assign(outer.begin() + c0, outer.begin() + std::min<uint64_t>(c0 + batch, outer.size()));
        }
        for (uint32_t ci = rank; ci < n_chunks; ci += world) {
            auto res = find_in_subset(be, chunks[ci], mask, fp_mask, tangent_mask, chunk_params(ci));
            chunk_record& r = local[ci / world];
            r.considered = res.second;
            if (!res.first.empty()) { /* at most one per chunk: the first sufficient hypothesis */
                const candidate_t refined = icp(be, res.first[0], p);
                r.found = 1;
                for (int k = 0; k < 16; ++k) r.pose[k] = refined.transform.m[k];
            }
        }
        ex.allgather(local.data(), per_rank, all.data());
        const float dt = p.max_icp_iterations == 0 ? p.dist_thres : 2.f * p.dist_thres; /* what icp() extracts with */
        for (uint32_t ci = 0; ci < n_chunks; ++ci) {
            const chunk_record& r = all[(size_t)(ci % world) * per_rank + ci / world];
            for (uint32_t k = 0, seen = 0; k < chunks[ci].size() && seen < r.considered; ++k, ++seen)
                fp_mask[chunks[ci][k]] = 1; /* :160-162 */
            if (!r.found) continue;
            pose_t T;
            for (int k = 0; k < 16; ++k) T.m[k] = r.pose[k];
            candidate_t c;
            be.extract(T, dt, c);
            cands.push_back(std::move(c));
        }
    } else if (!p.min_tangent_inliers &&
               detail::deferred_batches(be, detail::has_search_batches<Backend>{})) {
        /* the chunks' batches are gathered first (sampling on the host), scored in ONE call, then every chunk takes
         * its first batch with a sufficient hypothesis: the reference runs the chunks on concurrent threads */
        std::vector<std::vector<uint32_t>> chunks, batches;
        std::vector<uint32_t> cons_of, first_batch_of;
        {   /* every chunk's first points are known now: the backend may fetch their neighbourhoods across chunks */
            std::vector<uint32_t> live;
            for (uint32_t i : outer)
                if (!fp_mask[i] && !mask[i]) live.push_back(i);
            detail::announce_round(be, live, detail::has_announce_round<Backend>{});
        }
        for (uint64_t c0 = 0; c0 < outer.size(); c0 += batch)
            chunks.emplace_back(outer.begin() + c0, outer.begin() + std::min<uint64_t>(c0 + batch, outer.size()));
        /* The reference samples its chunks on concurrent threads (:146-166); each has its own generator and reads
         * shared state only. Here too when the backend has every neighbourhood view of the round at hand: the chunks'
         * sampling loops (80k draws per search at the configs[1] size) run on the worker pool and their batches are
         * joined in chunk order — the same batches as one chunk after the other. */
        const bool side_by_side = chunks.size() > 1 && p.workers > 1 && ::triplet_match::detail::worker_pool::instance().size() > 0 &&
                                  detail::round_views_ready(be, be.model_diameter(), detail::has_round_views_ready<Backend>{});
        if (side_by_side) {
            detail::timed whole(p.stats ? &p.stats->sample : nullptr);
            std::vector<std::vector<std::vector<uint32_t>>> chunk_batches(chunks.size());
            std::vector<stats_t> chunk_stats(chunks.size());
            std::vector<std::exception_ptr> failed(chunks.size());
            cons_of.assign(chunks.size(), 0u);
            ::triplet_match::detail::worker_pool::instance().run(chunks.size(), [&](size_t ci) {
                try {
                    params_t pl = chunk_params(ci);
                    pl.stats = p.stats ? &chunk_stats[ci] : nullptr;
                    cons_of[ci] = find_in_subset(be, chunks[ci], mask, fp_mask, tangent_mask, pl, &chunk_batches[ci]).second;
                } catch (...) {
                    failed[ci] = std::current_exception();
                }
            });
            for (auto& f : failed)
                if (f) std::rethrow_exception(f);
            for (size_t ci = 0; ci < chunks.size(); ++ci) {
                first_batch_of.push_back((uint32_t)batches.size());
                for (auto& b : chunk_batches[ci]) batches.push_back(std::move(b));
                if (p.stats) {
                    p.stats->firsts_sampled += chunk_stats[ci].firsts_sampled;
                    p.stats->draws += chunk_stats[ci].draws;
                    p.stats->neighbours += chunk_stats[ci].neighbours;
                }
            }
        } else {
            for (size_t ci = 0; ci < chunks.size(); ++ci) {
                first_batch_of.push_back((uint32_t)batches.size());
                auto res = find_in_subset(be, chunks[ci], mask, fp_mask, tangent_mask, chunk_params(ci), &batches);
                cons_of.push_back(res.second);
            }
        }
        first_batch_of.push_back((uint32_t)batches.size());
        detail::announce_round(be, std::vector<uint32_t>(), detail::has_announce_round<Backend>{});
        std::vector<char> found;
        std::vector<pose_t> poses;
        {
            TM_TIMED(generate_score);
            detail::run_batches(be, batches, be.model_diameter(), p.early_drop,
                                p.accept_prob * (float)n_model_all, found, poses, detail::has_search_batches<Backend>{});
        }
        std::vector<pose_t> starts; /* every chunk's first sufficient hypothesis, in chunk order */
        for (size_t ci = 0; ci < chunks.size(); ++ci) {
            for (uint32_t k = 0, seen = 0; k < chunks[ci].size() && seen < cons_of[ci]; ++k, ++seen)
                fp_mask[chunks[ci][k]] = 1; /* :160-162 */
            for (uint32_t b = first_batch_of[ci]; b < first_batch_of[ci + 1]; ++b)
                if (found[b]) {
                    starts.push_back(poses[b]);
                    if (tr) tr->chunks[ci].hit = true, tr->chunks[ci].hit_pose = poses[b];
                    break;
                }
        }
        /* icp_ of every candidate (:169-171). The candidates do not depend on one another, so a backend that can
         * refines them side by side in one call; icp_ itself only reads the candidate's transform (:375), so the
         * correspondence lists of finish_find (:361-364) are extracted once, for the refined pose. */
        std::vector<pose_t> refined;
        bool batched = false;
        if (p.max_icp_iterations > 0 && !starts.empty()) {
            TM_TIMED(icp);
            batched = detail::device_icp_batch(be, starts, 2.f * p.dist_thres, p.max_icp_iterations, refined,
                                               detail::has_icp_refine_batch<Backend>{});
        }
        if (batched && !tr) { /* finish_find and the arbitration below where the lists are, when the backend can */
            TM_TIMED(extract);
            std::vector<candidate_t> accepted;
            if (detail::extract_arbitrate(be, refined, 2.f * p.dist_thres, min_points, min_points_all, accepted, p.verbose,
                                          detail::has_extract_arbitrate<Backend>{}))
                return accepted;
        }
        if (batched) {
            TM_TIMED(extract);
            std::vector<candidate_t> got;
            detail::extract_all(be, refined, 2.f * p.dist_thres, got, detail::has_extract_batch<Backend>{});
            for (auto& c : got) cands.push_back(std::move(c));
        }
        for (size_t k = 0; k < starts.size() && !batched; ++k) {
            candidate_t c;
            {
                TM_TIMED(extract);
                be.extract(starts[k], p.dist_thres, c); /* :361-364 */
            }
            cands.push_back(icp(be, c, p)); /* :169-171 */
        }
    } else
    for (uint64_t c0 = 0; c0 < outer.size(); c0 += batch) { /* chunks, :156-166 (run one after another) */
        std::vector<uint32_t> chunk(outer.begin() + c0, outer.begin() + std::min<uint64_t>(c0 + batch, outer.size()));
        auto res = find_in_subset(be, chunk, mask, fp_mask, tangent_mask, chunk_params((size_t)(c0 / batch)));
        uint32_t cons = res.second;
        if (p.verbose)
            fprintf(stderr, "[search] chunk of %zu firsts: %u considered, %zu candidate(s)%s\n", chunk.size(), cons,
                    res.first.size(), res.first.empty() ? "" : " found");
        for (uint32_t k = 0, seen = 0; k < chunk.size() && seen < cons; ++k, ++seen) fp_mask[chunk[k]] = 1; /* :160-162 */
        for (auto& mt : res.first) cands.push_back(icp(be, mt, p)); /* :169-171 */
    }
    TM_TIMED(arbitration);
    auto to_trace = [](const std::vector<candidate_t>& from, std::vector<trace_match>& to) {
        for (const auto& c : from) to.push_back(trace_match{c.transform, c.scene_corrs, c.model_corrs, c.signed_score});
    };
    if (tr) to_trace(cands, tr->candidates);
    auto accepted = arbitrate(std::move(cands), be.scene_size(), tangent_mask, min_points, min_points_all, p.verbose);
    if (tr) to_trace(accepted, tr->accepted);
    return accepted;
}

/* vectors of find_all a caller may keep between searches over one scene (scene<Point> does): the two masks, and the
 * indices of the tangent points — valid as long as the tangent mask they were listed from is the one passed in (the
 * owner clears tangent_indices_valid when it builds another mask) */
struct find_all_workspace {
    std::vector<int32_t> mask, fp_mask;
    std::vector<uint32_t> tangent_indices;
    bool tangent_indices_valid = false;
    /* what the last search set in the two masks: mask_ at `mask_set` (the accepted matches' points), fp_mask_ only at
     * tangent points (first points are drawn from tangent_indices). While `sparse_reset` holds, the next search over
     * the same tangent mask clears those entries instead of 8 MB per million points. */
    std::vector<uint32_t> mask_set, added; /* `added`: a round's new indices (swapped with mask_set: neither is ever a fresh allocation) */
    bool sparse_reset = false;
    bool mask_tracked = true; /* false: a round's indices were not listed (the backend took them from its own rows): mask_ is cleared whole */
};

/* find_all_parallel, include/impl/scene.hpp:39-98. Returned transforms map scene -> model; the caller
 * inverts them (:92). */
template <class Backend, class Exchange>
inline std::vector<candidate_t> find_all(Backend& be, const std::vector<int32_t>& tangent_mask,
                                         const params_t& p, Exchange& ex,
                                         const std::vector<int32_t>* initial_mask = nullptr,
                                         find_all_workspace* keep = nullptr) {
    const uint32_t n = be.scene_size();
    /* mask_ starts empty in the reference (:41-44); the projected searches of one scene pass the points
     * earlier projectors have already explained. With a workspace the two masks are the caller's vectors, cleared
     * (4 MB each per million points: fresh vectors are fresh pages), and the list of tangent points is taken over
     * when the caller says it still belongs to this tangent mask. */
    find_all_workspace local;
    find_all_workspace& ws = keep ? *keep : local;
    std::vector<int32_t>&mask = ws.mask, &fp_mask = ws.fp_mask;
    std::vector<uint32_t>& tangent_indices = ws.tangent_indices;
    if (keep && ws.sparse_reset && ws.tangent_indices_valid && mask.size() == n && fp_mask.size() == n) {
        if (ws.mask_tracked)
            for (uint32_t idx : ws.mask_set) mask[idx] = 0;
        else
            mask.assign(n, 0);
        for (uint32_t idx : tangent_indices) fp_mask[idx] = 0;
    } else {
        mask.assign(n, 0);
        fp_mask.assign(n, 0);
    }
    ws.mask_tracked = true;
    ws.sparse_reset = false; /* until this search has ended in the ordinary way */
    ws.mask_set.clear();
    if (initial_mask) mask = *initial_mask;
    if (!ws.tangent_indices_valid) {
        tangent_indices.clear();
        for (uint32_t i = 0; i < n; ++i)
            if (tangent_mask[i]) tangent_indices.push_back(i);
        ws.tangent_indices_valid = keep != nullptr;
    }
    const uint32_t n_scene = (uint32_t)tangent_indices.size();
    TM_TIMED(total);
    {
        TM_TIMED(mask);
        const auto t_c = std::chrono::steady_clock::now();
        if (initial_mask)
            be.set_mask(mask);
        else
            detail::mask_clear(be, mask, detail::has_sparse_mask<Backend>{}); /* :41 */
        if (p.verbose)
            fprintf(stderr, "[mask] backend mask cleared %.3f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_c).count());
    }
    std::vector<candidate_t> results;
    std::vector<uint32_t>& added = ws.added;
    for (;;) {
        if (p.stats) ++p.stats->rounds;
        auto matches = find_parallel(be, tangent_indices, n_scene, mask, fp_mask, tangent_mask, p, ex);
        if (matches.empty()) break;
        TM_TIMED(mask);
        /* :88-90 for every accepted match (on this thread: the matches' indices interleave in the cloud's order, and
         * threads writing into the same cache lines cost more than they share). The backend's copy of mask_: from the lists it still holds when every match came out of its own arbitration
         * (nothing goes up, and the indices are not even listed: the next search clears mask_ whole), else the indices
         * go up */
        std::vector<uint32_t> rows;
        for (const auto& mt : matches)
            if (mt.device_row >= 0) rows.push_back((uint32_t)mt.device_row);
        const bool held = detail::has_mask_add_extracted<Backend>::value && rows.size() == matches.size();
        auto t_m = std::chrono::steady_clock::now();
        auto lap_m = [&](const char* what) {
            if (!p.verbose) return;
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[mask] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_m).count());
            t_m = now;
        };
        added.clear();
        if (held) {
            if (keep) ws.mask_tracked = false;
            for (const auto& mt : matches)
                for (uint32_t idx : mt.scene_corrs) mask[idx] = 1;
        } else {
            size_t total = 0;
            for (const auto& mt : matches) total += mt.scene_corrs.size();
            added.reserve(total);
            for (const auto& mt : matches) {
                for (uint32_t idx : mt.scene_corrs) mask[idx] = 1;
                added.insert(added.end(), mt.scene_corrs.begin(), mt.scene_corrs.end());
            }
        }
        lap_m("host mask");
        for (auto& mt : matches) {
            mt.device_row = -1; /* the rows belong to this round */
            results.push_back(std::move(mt));
        }
        if (held)
            detail::mask_add_extracted(be, rows, detail::has_mask_add_extracted<Backend>{});
        else
            detail::mask_add(be, mask, added, detail::has_sparse_mask<Backend>{});
        lap_m("backend mask");
        if (keep) {
            if (ws.mask_set.empty())
                ws.mask_set.swap(added);
            else
                ws.mask_set.insert(ws.mask_set.end(), added.begin(), added.end());
        }
    }
    ws.sparse_reset = keep != nullptr && initial_mask == nullptr;
    return results;
}

template <class Backend>
inline std::vector<candidate_t> find_all(Backend& be, const std::vector<int32_t>& tangent_mask,
                                         const params_t& p) {
    single_rank ex;
    return find_all(be, tangent_mask, p, ex);
}

}  // namespace search
}  // namespace triplet_match

#endif
