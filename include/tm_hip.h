/*
 * tm_hip.h — C ABI of the MI355X-native triplet-feature RANSAC hot path (libtm_hip.so).
 *
 * The reference (richard-vock/triplet_match) has no FFI: its boundary is the C++ template API
 * model<Point> / scene<Point> / *_traits<Point>. This header is the plain-C layer
 * those templates are re-hosted on (include/triplet_match/ headers); every entry point names the
 * reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions: opaque handles; int status return (0 = TM_OK, negative = error, text via
 * tm_last_error(), per thread); caller-owned buffers; no exceptions cross the ABI; one HIP stream per context.
 * Thread-safe per handle: every entry point that takes a context, or a model / scene / cloud / communicator created
 * from it, holds that context's mutex for the call, so calls from several threads on one context serialise (the
 * reference's workers call model::query / voxel_query / radius_search concurrently, include/impl/scene.hpp:146-166);
 * contexts are independent of each other. Destroying a handle while another thread is inside a call on it is the
 * caller's error, as with any C handle. The pure functions (tm_feature, tm_murmur3_32, tm_traits_* ...) take no lock.
 * Pointers are HOST pointers unless the parameter name starts with `d_` (device, on the ctx device).
 *
 * Layouts:
 *   surfel  48 B = 12 x f32  [x y z 1 | nx ny nz 0 | rgba tx ty tz]  (pcl::PointSurfel; the tangent
 *           aliases data_c[1..3], include/common:62-70)
 *   pose    16 x f32 column-major 4x4, maps scene -> model (include/impl/scene.hpp:538-567)
 *   feature 4 x f32; key 4 x u32; table entry (u32 i, u32 j) = model point indices
 */
#ifndef TM_HIP_H_
#define TM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TM_OK 0
#define TM_ERR_INVALID -1         /* bad argument */
#define TM_ERR_NOT_INITIALIZED -2 /* reference: std::runtime_error "Cannot query uninitialized model" */
#define TM_ERR_HIP -3             /* HIP runtime error, text in tm_last_error() */
#define TM_ERR_EMPTY -4           /* no valid pair (reference crashes, include/impl/model.hpp:151) */
#define TM_ERR_UNSUPPORTED -5
#define TM_ERR_NO_DEVICE -6       /* no gfx950 device / HIP not usable: there is no CPU fallback */
#define TM_ERR_CAPACITY -7        /* output buffer too small; *count holds the required size */

#define TM_TRAITS_IDENTITY 0 /* include/identity_traits */
#define TM_TRAITS_PLANE 1    /* include/plane_traits */
#define TM_TRAITS_PLANE2 2   /* include/plane2_traits */
#define TM_TRAITS_CYLINDER 3 /* include/cylinder_traits, opencl/cylinder.cl */
#define TM_STATE_FLOATS 37   /* traits state: radius, threshold, origin[3], g2l[16], l2g[16] */

#define TM_QUERY_LIMIT_DEFAULT 200 /* include/impl/scene.hpp:19 */
#define TM_VOXEL_MARGIN 5          /* include/impl/model.hpp:48 */

typedef struct tm_ctx tm_ctx;
typedef struct tm_model tm_model;
typedef struct tm_scene tm_scene;

const char* tm_last_error(void);
const char* tm_version(void);

/* ---- context ---- */
int tm_ctx_create(int device_id, tm_ctx** ctx);
int tm_ctx_destroy(tm_ctx* ctx);
int tm_ctx_synchronize(tm_ctx* ctx);
void* tm_ctx_stream(tm_ctx* ctx); /* hipStream_t all kernels of this ctx are launched on */
/* device scratch owned by the ctx (for callers without their own allocator) */
int tm_device_alloc(tm_ctx* ctx, uint64_t bytes, void** d_ptr);
int tm_device_free(tm_ctx* ctx, void* d_ptr);
/* The context's cache of released device blocks (work buffers of the calls, the arrays of destroyed scenes and
 * models): a block goes to the next request of its size class instead of hipFree + hipMalloc, in stream order on the
 * context's one stream. Budget = option "pool_mib" (default 4096 MiB, 0 = no cache; a smaller value trims at once; a release past the budget trims to three quarters of it).
 * cached_bytes = held right now, reused / allocated = requests served from the cache / by hipMalloc so far. The
 * reference has no counterpart (host vectors, include/impl/scene.hpp:424-430). */
int tm_ctx_pool_stats(const tm_ctx* ctx, uint64_t* cached_bytes, uint64_t* reused, uint64_t* allocated);
int tm_memcpy_h2d(tm_ctx* ctx, void* d_dst, const void* src, uint64_t bytes);
int tm_memcpy_d2h(tm_ctx* ctx, void* dst, const void* d_src, uint64_t bytes);
/* Switches of a context (cross-check kernels, timing comparisons; none changes a result). Every switch NAME is
 * read once from the environment variable TM_<NAME IN UPPER CASE> inside tm_ctx_create; afterwards only these
 * calls change it — no call of the path reads the environment. Names: debug_stats, debug_times, debug_ed,
 * score_grouped, ball_ungrouped, score_item_wgs, score_slot_mib, item_sort, near_metric, voxel_bruteforce,
 * voxel_packet, voxel_seed, curv_groups, model_host_build, ed_one_kernel, ed_narrow, ed_bruteforce, ed_unfused,
 * search_segment, search_per_range, search_step_check, ed_wave_rows, ed_prefix_list, ed_hist, ed_ball_list,
 * ed_prefix_group, ed_bound_first, score_off32, pool_mib (what each selects: the comment at its field in
 * csrc/tm_internal.h, struct options).
 * "ablate" (stage ablation: WRONG results, timing only) is refused with TM_ERR_UNSUPPORTED unless the library
 * was built with -DTM_TIMING_EXPERIMENTS; tm_version() of such a build says so. Unknown name / value out of
 * range: TM_ERR_INVALID. */
int tm_ctx_set_option(tm_ctx* ctx, const char* name, int value);
int tm_ctx_get_option(const tm_ctx* ctx, const char* name, int* value);
/* A stopwatch on the context's stream: HIP events around everything queued between the two calls (_end waits for
 * it). For measuring device-resident calls such as tm_score_hypotheses_early_drop_ball_dev. */
int tm_ctx_time_begin(tm_ctx* ctx);
int tm_ctx_time_end(tm_ctx* ctx, float* ms);
/* HIP-event timing of the scoring kernel launches on the ctx stream (bench.py roofline). While timing is
 * on, every scoring launch records an event pair and one row of device counters; nothing is copied to the
 * host and nothing synchronises until one of the two read-out calls below (they synchronise the stream).
 * tm_ctx_enable_timing(ctx, 1) opens a new measurement window (the totals start from zero); (ctx, 2) opens one with the
 * events only: the stage counters read zero, and the grouped kernel runs the instantiation without their bookkeeping —
 * the one every launch outside a measurement window runs (three to four VALU instructions per evaluation chain). */
int tm_ctx_enable_timing(tm_ctx* ctx, int on);
/* the most recent timed launch */
int tm_ctx_last_score_timing(tm_ctx* ctx, float* ms, uint64_t* evaluations, uint64_t* passing,
                             uint64_t* hypotheses);
/* sums over the window's launches (the last 256 at most): totals6 = evaluations, distance-passing,
 * in-grid, near (in-grid and brick code within reach), hypotheses, 512-point scene chunks staged into LDS
 * (grouped kernel; 0 for the others) */
int tm_ctx_score_timing_totals(tm_ctx* ctx, uint32_t* launches, float* total_ms, uint64_t totals6[6]);

/* ---- scalar descriptor functions, host side ----
 * replace: feature() include/impl/feature.hpp:16-34; discretize() x2 src/discretize.cpp:20-30;
 * discretize_feature() include/impl/feature.hpp:38-46; valid() :50-88; valid_bounds() :91-114;
 * std::hash<Matrix<u32,Dim,1>> = murmur<Dim> include/impl/discretize.hpp:5-56 */
uint32_t tm_murmur3_32(const uint32_t* words, int n_words, uint32_t seed);
uint32_t tm_hash_key(const uint32_t* key, int dim); /* seed 42 */
uint32_t tm_discretize_range(float value, float min_value, float range_value, uint32_t steps);
uint32_t tm_discretize_step(float value, float step_size);
float    tm_atan2f(float y, float x);
void     tm_feature(const float* surfel0, const float* surfel1, float feature[4]);
void     tm_discretize_feature(const float feature[4], const float bmin[4], const float bmax[4],
                               float distance_step_count, float angle_step, uint32_t key[4]);
int      tm_valid(const float feature[4], const float bmin[4], const float bmax[4]);
void     tm_valid_bounds(const float bmin[4], const float bmax[4], float min_rel_dist,
                         float max_rel_dist, float out_min[4], float out_max[4]);
/* replaces scene::impl::base_transform_ include/impl/scene.hpp:538-567 */
void     tm_base_transform(const float src_i[3], const float src_j[3], const float src_t[3],
                           const float tgt_i[3], const float tgt_j[3], const float tgt_t[3],
                           float pose16[16]);
uint32_t tm_early_drop_upper(uint32_t tried, uint32_t subset_size, uint32_t corrs);

/* batched descriptor kernel (device): features + keys of n (a,b) index pairs into a surfel array.
 * bounds may be NULL (then keys are not produced). Parity entry point for "bit-exact triplet keys". */
int tm_features_batch(tm_ctx* ctx, const float* surfels, uint32_t n_points, const uint32_t* ab_pairs,
                      uint32_t n_pairs, const float bmin[4], const float bmax[4],
                      float distance_step_count, float angle_step, float* features4,
                      uint32_t* keys4, uint32_t* hashes);

/* cloud->resolution(): mean distance to the nearest OTHER point (reference
 * include/impl/pointcloud.hpp:66-82: kd-tree nearestKSearch(pnt, 2) -> sqrtf(ds[1]), then the running
 * mean accum + (val - accum) / (++i) in f32 over the points in order, include/common:104-115).
 * The 2-NN distances come from a kd-tree kernel on the GPU (exact), the order-dependent running mean is
 * taken on the host. nn_dist (optional, n floats) receives the per-point distances. */
int tm_cloud_resolution(tm_ctx* ctx, const float* surfels, uint32_t n, float* resolution, float* nn_dist);

/* cloud->curvature(k, idx) for every point (reference include/impl/pointcloud.hpp:3-44,200-204):
 * k nearest neighbours including the point itself (kd-tree kernel; order = distance ascending, ties by
 * index — FLANN's tie order is unpinned), normals projected into the tangent plane of the point's own
 * normal, running-mean centroid, 3x3 covariance, eigenvalues. pcl::eigen33 is third-party: defined here
 * as a 10-sweep cyclic Jacobi iteration in f32. curv5[i] = (pc_max_dir xyz, pc_min, pc_max) with
 * pc_min = middle, pc_max = largest eigenvalue, both / k. The reference's tangent masks are
 * |tangent| > 0.7 && pc_min / pc_max < 0.2 (include/impl/scene.hpp:50, model.hpp:96-99). */
int tm_cloud_curvature(tm_ctx* ctx, const float* surfels, uint32_t n, uint32_t k, float* curv5);
/* The same for the listed points only (row q of curv5 = point indices[q]; the neighbours still come from the whole
 * cloud). The reference computes the curvature of every point (include/impl/scene.hpp:48-49, model.hpp:68-71) but
 * reads it only where the tangent test passes (:50, :96-99) — 4 % of the points at the configs[1] sizes. */
int tm_cloud_curvature_subset(tm_ctx* ctx, const float* surfels, uint32_t n, uint32_t k, const uint32_t* indices,
                              uint32_t n_indices, float* curv5);
/* The same for a scene that is already on the device (mask_ is ignored: the neighbours come from every finite point):
 * no second upload of the cloud and no tree built on the host — the scene's points are in curve order already, and a
 * complete binary tree over that order with true bounding boxes is a few launches (a 1M-point scene: 70 ms of host
 * preparation for a 3 ms kernel otherwise). Same neighbours, same arithmetic, same rows as tm_cloud_curvature_subset.
 * With tm_scene_set_tangent_mask this lets a caller create the scene first, ask the curvature of the candidate
 * points, and hand in tangent_mask_ (include/impl/scene.hpp:46-54) afterwards. */
int tm_scene_curvature_subset(tm_scene* scene, uint32_t k, const uint32_t* indices, uint32_t n_idx, float* curv5);
/* replaces tangent_mask_ of an existing scene (the sorted order and the cloud stay) */
int tm_scene_set_tangent_mask(tm_scene* scene, const int32_t* tangent_mask);
/* tangent_mask_ built where the cloud already is (include/impl/scene.hpp:46-54 in one call on the device):
 *   mask[i] = |tangent(p_i)| > min_norm  [&& ok[i] != 0]  && (k == 0 || pc_min(i) / pc_max(i) < max_ratio)
 * (reference: 0.7, 0.2, k = 30; ok may be null), the curvature as tm_scene_curvature_subset computes it for exactly the
 * points that pass the first tests. The mask becomes the scene's (as tm_scene_set_tangent_mask) and is handed back:
 * tangent_mask (n entries), the candidates in ascending order with their curvature rows (cap entries / 5 * cap floats;
 * TM_ERR_CAPACITY with *n_candidates set if more were found), each optional. Same mask as the three calls
 * and the host loop it replaces, bit for bit. */
int tm_scene_auto_tangent_mask(tm_scene* scene, uint32_t k, float min_norm, float max_ratio, const unsigned char* ok,
                               int32_t* tangent_mask, uint32_t* candidates, float* curv5, uint32_t cap,
                               uint32_t* n_candidates, uint32_t* n_tangent);

/* ---- neighbour queries of pointcloud<Point> (include/pointcloud:38-63, include/impl/pointcloud.hpp:100-210) ----
 * tm_cloud = the cloud's search structure: kdtree() / build_tree_ (:100-136, :224-240), over all finite points or
 * over `subset` (set_indices, include/pointcloud:95-103). The reference builds it lazily under a mutex and its
 * workers query it concurrently; the handle's queries are serialised by a mutex of its own, so a handle may be
 * shared by threads. Results are ordered by (squared distance, index) ascending with squared distance =
 * (dx*dx + dy*dy) + dz*dz in f32, d = point - query (FLANN's tie order is unpinned; this is the order of
 * tm_cloud_curvature and of the oracle); distances are SQUARED, as PCL's nearestKSearch / radiusSearch return. */
typedef struct tm_cloud tm_cloud;
int tm_cloud_create(tm_ctx* ctx, const float* surfels, uint32_t n, const uint32_t* subset, uint32_t n_subset,
                    tm_cloud** cloud);
int tm_cloud_destroy(tm_cloud* cloud);
uint32_t tm_cloud_tree_size(const tm_cloud* cloud); /* points in the tree (finite, in the subset) */
/* knn_inclusive(k, point) for a batch (include/impl/pointcloud.hpp:138-146): queries_xyz = n_queries x 3 floats;
 * indices / sq_dists = n_queries x k, rows padded with -1 / +inf when the tree holds fewer than k points;
 * counts (optional) = entries per row. k <= 32: one kd-tree launch for the batch; larger k: one pass over the
 * points per query. knn_exclusive(k, .) = the row of k + 1 without its first entry (:154-160). */
int tm_cloud_knn(tm_cloud* cloud, const float* queries_xyz, uint32_t n_queries, uint32_t k, int32_t* indices,
                 float* sq_dists, uint32_t* counts);
/* radius_search_inclusive(r, point) (:168-176): members = sq dist <= r*r; *count = total, the first
 * min(cap, total) are written (TM_ERR_CAPACITY if cap is too small; indices and sq_dists may both be NULL to
 * count). radius_search_exclusive = the same without its first entry (:184-190). */
int tm_cloud_radius_search(tm_cloud* cloud, const float query_xyz[3], float radius, int32_t* indices,
                           float* sq_dists, uint32_t cap, uint32_t* count);
/* principal_curvatures(cloud, idx, neighbours) (:3-44) over an explicit neighbour list, as curvature(float r, idx)
 * feeds it with radius_search_inclusive(r, idx).first (:206-209); curv5 as tm_cloud_curvature */
int tm_cloud_curvature_of(const tm_cloud* cloud, uint32_t idx, const int32_t* neighbours, uint32_t n_neighbours,
                          float curv5[5]);

/* ---- model: replaces model<Point> (include/model:15-66, include/impl/model.hpp) ---- */
/* model(cloud, discretization_params); `resolution` = cloud->resolution() (mean 1-NN distance) */
int tm_model_create(tm_ctx* ctx, const float* surfels, uint32_t n, float resolution,
                    float distance_step_count, float angle_step, tm_model** model);
int tm_model_destroy(tm_model* model);
/* init(subset, ...) of the reference takes the bounding box, hence diameter(), the distance window of the
 * pairs and the voxel grid, over the finite points of the subset it is handed
 * (include/impl/model.hpp:17-38); an empty subset stands for the whole cloud. Call before
 * tm_model_build_voxels when the caller's subset is not empty; (NULL, 0) restores the default.
 * The voxel NN itself always searches the whole cloud (:88). */
int tm_model_set_grid_subset(tm_model* model, const uint32_t* subset, uint32_t n_subset);
/* init() part A, include/impl/model.hpp:34-66,81-94: bbox, diameter, to_voxel, voxel NN grid */
int tm_model_build_voxels(tm_model* model);
/* init() part B, include/impl/model.hpp:96-149: `subset` is subset_ after the tangent/curvature
 * filter (:96-99). Builds feature bounds and the key -> (i,j) table. */
int tm_model_build_pairs(tm_model* model, const uint32_t* subset, uint32_t n_subset,
                         float min_diameter_factor, float max_diameter_factor);
/* Declares the model to be a projector's uvw cloud (built from tm_traits_project_cloud output): the
 * query <-> model-point distance of the scoring kernels becomes that traits' intrinsic_distance
 * (plane2: 2-D; cylinder: wrap-around in u with `radius`); remnant of the reference's
 * model<Proj,Point>::voxel_distance (include/impl/model.hpp:194-201). Default: identity (3-D L2). */
int tm_model_set_metric(tm_model* model, int traits_kind, float radius);
int tm_model_get_info(const tm_model* model, float* diameter, uint32_t* cloud_size,
                      uint32_t* point_count /* tangent-subset size, model.hpp:219-221 */,
                      float* resolution);
/* extents(), voxel_transform(), margin(): include/model:49-55 */
int tm_model_get_grid(const tm_model* model, int extents[3], float to_voxel16[16]);
int tm_model_download_voxels(const tm_model* model, uint32_t* voxels, uint64_t cap, uint64_t* count);
/* feature_bounds(): include/model:61 */
int tm_model_get_feature_bounds(const tm_model* model, float bmin[4], float bmax[4]);
/* number of table entries and of pass-1 valid pairs (pair_count() is uninitialised in the
 * reference, include/impl/model.hpp:223-225) */
int tm_model_get_pair_counts(const tm_model* model, uint64_t* entries, uint64_t* valid_pairs,
                             uint32_t* distinct_keys);
/* table dump for parity: distinct keys sorted lexicographically, CSR offsets (n_keys+1), entries in
 * canonical query order (libstdc++ equal_range order = reverse insertion) */
int tm_model_download_table(const tm_model* model, uint32_t* keys4, uint32_t* offsets,
                            uint32_t key_cap, uint32_t* pairs_ij, uint64_t pair_cap);
/* query(feature): include/impl/model.hpp:169-178. Writes min(cap, total) entries in canonical
 * order, *count_total = equal_range size. TM_ERR_NOT_INITIALIZED before build_pairs. */
int tm_model_query(const tm_model* model, const float feature[4], uint32_t* pairs_ij, uint32_t cap,
                   uint32_t* count_total);
int tm_model_query_key(const tm_model* model, const uint32_t key[4], uint32_t* pairs_ij,
                       uint32_t cap, uint32_t* count_total);
/* voxel_query(pos): include/impl/model.hpp:180-192. *found = 0 for nullopt. */
int tm_model_voxel_query(const tm_model* model, const float pos4[4], uint32_t* index, int* found);

/* ---- scene: replaces scene<Point> (include/scene:10-59, include/impl/scene.hpp) ---- */
/* scene(cloud) + tangent_mask_ (include/impl/scene.hpp:42,50-53). tangent_mask may be null: no tangent point yet
 * (tm_scene_auto_tangent_mask builds the reference's mask on the device, tm_scene_set_tangent_mask takes one). The
 * cloud is copied to the device only; host-side helpers that read rows (tm_scene_download, tm_pair_is_valid_sample,
 * tm_scene_tangent_points) fetch a host mirror from there on first use. */
int tm_scene_create(tm_ctx* ctx, const float* surfels, uint32_t n, const int32_t* tangent_mask,
                    tm_scene** scene);
/* The same for a cloud that is already on the context's device (a sensor pipeline's output): d_surfels = n x 12
 * floats, d_tangent_mask = n x int32 or null; both are copied device to device on the context's stream — the caller's
 * own work on them must be complete — and nothing crosses PCIe but the 4 n bytes of the mask for the host helpers. */
int tm_scene_create_dev(tm_ctx* ctx, const float* d_surfels, uint32_t n, const int32_t* d_tangent_mask,
                        tm_scene** scene);
int tm_scene_destroy(tm_scene* scene);
/* mask_ (include/impl/scene.hpp:41,87-90); NULL clears */
int tm_scene_set_mask(tm_scene* scene, const int32_t* mask);
/* mask_[idx] = 1 for the n listed points on top of the mask the scene holds (the loop of include/impl/scene.hpp:87-90
 * over an accepted match's scene_corrs): same state as tm_scene_set_mask with the updated mask, without moving 4 bytes
 * per scene point up and down for a few correspondences' worth of change. */
int tm_scene_mask_add(tm_scene* scene, const uint32_t* indices, uint32_t n);
/* ... the scene points of rows `rows` of the LAST tm_extract_correspondences_batch / tm_extract_arbitrate_batch call on
 * this scene, whose lists are still on the device (the accepted matches of a find_parallel round, :87-90) */
int tm_scene_mask_add_extracted(tm_scene* scene, const uint32_t* rows, uint32_t n_rows);
uint32_t tm_scene_size(const tm_scene* scene);

/* nn = cloud_->radius_search_inclusive(radius, p1) as used at include/impl/scene.hpp:273
 * (include/impl/pointcloud.hpp:172-183; PCL/FLANN leave the result order open): the indices j with
 * |p_j - p_center|^2 <= radius^2 (fp32, ((dx*dx + dy*dy) + dz*dz)), ascending, the centre included.
 * Membership is computed on the device; *count = total, out receives min(cap, total) entries
 * (out may be NULL to count only). */
int tm_scene_radius_subset(tm_scene* scene, uint32_t center_idx, float radius, int32_t* out, uint32_t cap,
                           uint32_t* count);

/* Entries ranks[0..n_ranks) of the list the LAST tm_scene_radius_subset call on this scene described (that call may
 * have been made with out == NULL to count only): for a caller that needs a few random members of a large
 * neighbourhood — the traits search's draw of its second and third sample point — and not 100,000 indices. */
int tm_scene_radius_select(const tm_scene* scene, const uint32_t* ranks, uint32_t n_ranks, int32_t* out);

/* The same for several centres in one call, with the index lists built on the device (membership bits ->
 * per-word prefix counts -> ordered scatter): counts[k] = size of the neighbourhood of centers[k],
 * lists[k * stride + j] its j-th index (ascending). lists may be NULL to get the counts only; stride must
 * cover the largest neighbourhood (TM_ERR_CAPACITY otherwise). */
int tm_scene_radius_lists(tm_scene* scene, const uint32_t* centers, uint32_t n_centers, float radius,
                          int32_t* lists, uint32_t stride, uint32_t* counts);

/* What the sampling loop of find_in_subset (include/impl/scene.hpp:273-302) uses of those neighbourhoods: the size
 * of nn (inner_bound, :276-281; the range of the shuffle's draws, :285) and its members with tangent_mask_ set —
 * every other member fails the first filter of :290 whatever its index is. counts[k] = size of the neighbourhood
 * of centers[k]; t_counts[k] = its tangent members; pairs = their (rank, index), 2 x uint32 each, slot after slot
 * (slot k starts at the sum of the t_counts before it), each slot in ascending index order; rank = the member's
 * position in that neighbourhood's ascending order (the order tm_scene_radius_lists hands out). *n_pairs = total.
 * TM_ERR_CAPACITY (counts, t_counts and *n_pairs valid) when cap_pairs is too small. A 60,000-point neighbourhood
 * with 4 % tangent points travels as 19 KB instead of 240 KB. */
int tm_scene_radius_tangent_lists(tm_scene* scene, const uint32_t* centers, uint32_t n_centers, float radius,
                                  uint32_t* counts, uint32_t* t_counts, uint32_t* pairs, uint32_t cap_pairs,
                                  uint32_t* n_pairs);

/* host-side predicate of the sample filters of find_in_subset (include/impl/scene.hpp:290-302):
 * tangent_mask[j], masks, i != j, squared-distance window, collinearity, feature validity. Same
 * arithmetic as the device kernel; lets host orchestration count "valid samples" (:305,350). */
/* Returns 1 / 0, or a negative TM_ERR_* when the scene's host mirror (fetched from the device on the first call of a
 * host-side helper: 48 bytes per point and one stream synchronise) could not be read. */
int tm_pair_is_valid_sample(const tm_scene* scene, const tm_model* model, uint32_t i, uint32_t j,
                            float min_diameter_factor, float max_diameter_factor);
/* The same filters (:293-302: distance window, collinearity, feature validity) on two surfels the caller holds, with
 * the model's diameter and feature bounds (tm_model_get_info, tm_model_get_feature_bounds): a pure function — no
 * handle, no lock — for a sampling loop that runs on several host threads; the mask tests of :290 are the caller's.
 * tm_pair_is_valid_sample = mask tests + this on the scene's own rows. */
int tm_pair_valid_raw(const float* surfel_i, const float* surfel_j, float model_diameter, float min_diameter_factor,
                      float max_diameter_factor, const float bmin[4], const float bmax[4]);

/* hypothesis generation: inner body of find_in_subset, include/impl/scene.hpp:288-319, for an
 * ordered batch of scene (i,j) sample pairs: filters -> feature -> valid -> query (first
 * query_limit entries) -> base_transform_ -> optional force_up. Output compacted in visit order:
 * pose (16 f32) and provenance (pair index, m_i, m_j, entry rank). If more than `cap` hypotheses
 * exist the first `cap` are written and *count = cap (total in *count_total). */
int tm_generate_hypotheses(tm_scene* scene, tm_model* model, const uint32_t* ij_pairs,
                           uint32_t n_pairs, float min_diameter_factor, float max_diameter_factor,
                           int force_up, int query_limit, float* poses16, uint32_t* provenance4,
                           uint32_t cap, uint32_t* count, uint32_t* count_total);
/* same with device-resident input/output (bench / pipelines): d_ij_pairs, d_poses16, d_provenance4 */
int tm_generate_hypotheses_dev(tm_scene* scene, tm_model* model, const uint32_t* d_ij_pairs,
                               uint32_t n_pairs, float min_diameter_factor,
                               float max_diameter_factor, int force_up, int query_limit,
                               float* d_poses16, uint32_t* d_provenance4, uint32_t cap,
                               uint32_t* count, uint32_t* count_total);

/* inlier scoring: project_ (include/impl/scene.hpp:411-510) for a batch of poses, full pass (no
 * early drop), over `subset` (scene indices, NULL = all points = finish_find :100-106).
 * counts[h] = number of correspondences; scores[h] = sum |ref . ref_n| / |model cloud| (:509).
 * mode: TM_SCORE_AUTO picks the grid-culled kernel; TM_SCORE_BRUTEFORCE visits every point. */
#define TM_SCORE_AUTO 0
#define TM_SCORE_BRUTEFORCE 1
#define TM_SCORE_CULLED 2
int tm_score_hypotheses(tm_scene* scene, tm_model* model, int traits_kind, const float* poses16,
                        uint32_t n_hyp, const int32_t* subset, uint32_t n_subset, float dist_thres,
                        int mode, uint32_t* counts, double* scores);
int tm_score_hypotheses_dev(tm_scene* scene, tm_model* model, int traits_kind,
                            const float* d_poses16, uint32_t n_hyp, float dist_thres, int mode,
                            uint32_t* d_counts, double* d_scores);
/* Neighbourhood-restricted scoring for a batch (find_in_subset scores every hypothesis on
 * nn = radius_search_inclusive(m.diameter(), p1), include/impl/scene.hpp:273,326): hypothesis h is scored
 * on the scene points whose squared distance to scene point centers[h] is <= radius*radius (same set as
 * that radius search; masked points excluded), full pass. One launch for the whole batch: the ball test
 * runs inside the culled kernel (tile level conservatively, point level exactly). */
int tm_score_hypotheses_ball(tm_scene* scene, tm_model* model, const float* poses16, uint32_t n_hyp,
                             const uint32_t* centers, float radius, float dist_thres, uint32_t* counts,
                             double* scores);
int tm_score_hypotheses_ball_dev(tm_scene* scene, tm_model* model, const float* d_poses16, uint32_t n_hyp,
                                 const uint32_t* d_centers, float radius, float dist_thres,
                                 uint32_t* d_counts, double* d_scores);

/* The same full pass (project_ without early drop, include/impl/scene.hpp:411-510) over a LIST of scene points,
 * each listed at most once (TM_ERR_INVALID otherwise; at most 65,535 poses per call): one thread per (pose,
 * point) straight from the list instead of a walk over the whole scene — the form for short lists (a tangent
 * subset, a neighbourhood). counts / scores equal tm_score_hypotheses with the same points as subset. */
int tm_score_hypotheses_on_points(tm_scene* scene, tm_model* model, const float* poses16, uint32_t n_hyp,
                                  const int32_t* points, uint32_t n_points, float dist_thres, uint32_t* counts,
                                  double* scores);

/* early-drop rule of project_ (include/impl/scene.hpp:422-426,492-506) evaluated exactly for a
 * batch: dropped_at[h] = checkpoint index (0..17) or -1, counts/scores/saved as project_ returns */
int tm_score_hypotheses_early_drop(tm_scene* scene, tm_model* model, const float* poses16,
                                   uint32_t n_hyp, const int32_t* subset, uint32_t n_subset,
                                   float dist_thres, float accept_prob, uint32_t* counts,
                                   double* scores, int32_t* dropped_at, uint32_t* saved);

/* The same rule for a batch whose hypotheses each scan their own neighbourhood: hypothesis h scans
 * nn = radius_search_inclusive(radius, centers[h]) in ascending index order — the subset find_in_subset
 * hands to project_ for the first sample point centers[h] (include/impl/scene.hpp:273,326). Lets the host
 * loop push the hypotheses of several first points through one call. */
int tm_score_hypotheses_early_drop_ball(tm_scene* scene, tm_model* model, const float* poses16,
                                        uint32_t n_hyp, const uint32_t* centers, float radius,
                                        float dist_thres, float accept_prob, uint32_t* counts,
                                        double* scores, int32_t* dropped_at, uint32_t* saved);

/* The same with the poses already on the device and the results left there: only the centre indices come from the
 * host (4 bytes per hypothesis; their distinct values name the neighbourhoods to build). No synchronisation at the
 * end: the results are complete when the context's stream has drained. */
int tm_score_hypotheses_early_drop_ball_dev(tm_scene* scene, tm_model* model, const float* d_poses16, uint32_t n_hyp,
                                            const uint32_t* centers, float radius, float dist_thres, float accept_prob,
                                            uint32_t* d_counts, double* d_scores, int32_t* d_dropped_at,
                                            uint32_t* d_saved);

/* One batch of find_in_subset's inner loop in one call (include/impl/scene.hpp:284-350 for up to 16 first
 * points): the hypotheses of the valid sample pairs (i, j) are generated in visit order (pair after pair, the
 * pair's table entries in stored order, at most query_limit each; query_limit > 0), every hypothesis is scored
 * on nn = radius_search_inclusive(radius, i) of its own first point (with the early-drop rule when early_drop),
 * and the first hypothesis in visit order with score > 0 and (float)count > sufficient is returned (:330-338).
 * Same results as tm_generate_hypotheses + tm_score_hypotheses_[early_drop_]ball + a host scan, without the
 * poses and the per-hypothesis results crossing PCIe: 8 bytes per pair go up, one 88-byte record comes back. */
typedef struct tm_batch_result {
    int32_t index;         /* position of that hypothesis in the batch, -1 = none is sufficient */
    uint32_t n_hypotheses; /* hypotheses generated from the pairs */
    uint32_t count;        /* its correspondences */
    int32_t dropped_at;    /* its early-drop checkpoint, -1 = scanned to the end */
    double score;          /* as tm_score_hypotheses_[early_drop_]ball reports it */
    float pose[16];
} tm_batch_result; /* 88 bytes */
int tm_search_batch(tm_scene* scene, tm_model* model, const uint32_t* ij_pairs, uint32_t n_pairs,
                    float min_diameter_factor, float max_diameter_factor, int force_up, int query_limit,
                    float radius, float dist_thres, float accept_prob, int early_drop, float sufficient,
                    tm_batch_result* result);
/* Several independent batches in one call (the chunks of one find_parallel round, include/impl/scene.hpp:156-166:
 * the reference runs them on concurrent threads): pairs batch_offsets[b] .. batch_offsets[b + 1] belong to batch
 * b (offsets ascend from 0 to n_pairs), results[b] is what tm_search_batch returns for those pairs alone. The
 * hypotheses are generated in one go, the neighbourhoods of all first points built together, every batch runs
 * its own segments with its own stop flag, and nothing is read back before the end. */
int tm_search_batches(tm_scene* scene, tm_model* model, const uint32_t* ij_pairs, uint32_t n_pairs,
                      const uint32_t* batch_offsets, uint32_t n_batches, float min_diameter_factor,
                      float max_diameter_factor, int force_up, int query_limit, float radius, float dist_thres,
                      float accept_prob, int early_drop, float sufficient, tm_batch_result* results);

/* finish_find (include/impl/scene.hpp:100-106): correspondences of one pose over all points,
 * ascending scene index; *score normalised by |model cloud| */
int tm_extract_correspondences(tm_scene* scene, tm_model* model, const float pose16[16],
                               float dist_thres, uint32_t* scene_corrs, uint32_t* model_corrs,
                               uint32_t cap, uint32_t* count, double* score);

/* finish_find for n poses in one call (the reference calls it once per candidate, include/impl/scene.hpp:361-364, and
 * at the end of icp_, :401): the passes are queued back to back, ONE read-back of the counts and one of the lists
 * (through pinned memory). scene_corrs / model_corrs = n rows of `stride` entries, counts[k] of row k valid, each
 * row what tm_extract_correspondences returns for that pose alone. TM_ERR_CAPACITY (counts valid) when stride is
 * smaller than a list. */
int tm_extract_correspondences_batch(tm_scene* scene, tm_model* model, const float* poses16, uint32_t n,
                                     float dist_thres, uint32_t* scene_corrs, uint32_t* model_corrs, uint32_t stride,
                                     uint32_t* counts, double* scores);

/* The same, and the arbitration of find_parallel over the n candidates these poses are (include/impl/scene.hpp:
 * 193-237) while their lists are still on the device: sorted by number of correspondences (descending, stable), a
 * candidate keeps the scene points no ACCEPTED earlier candidate holds and is accepted when at least min_points of
 * them are tangent points (the scene's tangent mask) and at least min_points_all in total (:114-115, :227).
 * verdict[k] = 1 accepted, 0 rejected (its rows are not written), kept[k] / tangent_kept[k] = its two counts (either
 * may be null). verdict = -1 for EVERY k when the host has to decide: an accepted candidate would lose points to an
 * earlier one (its lists need trimming), or n > 254 — all rows are then written, as by the call above. */
int tm_extract_arbitrate_batch(tm_scene* scene, tm_model* model, const float* poses16, uint32_t n, float dist_thres,
                               uint32_t min_points, uint32_t min_points_all, uint32_t* scene_corrs,
                               uint32_t* model_corrs, uint32_t stride, uint32_t* counts, double* scores,
                               int32_t* verdict, uint32_t* kept, uint32_t* tangent_kept);

/* One ICP accumulation pass (reference icp_, include/impl/scene.hpp:385-393 + icp_correlation,
 * opencl/icp.cl:55-86): finds the correspondences of `pose16` exactly like finish_find (all points) and
 * returns their number and 15 sums over them — sum s (3), sum m (3), sum s_a*m_b (9, index 3*a+b) with s
 * the scene point and m its model point — accumulated on the device in 2^-32 fixed point (each term
 * (long long)(x * 2^32), integer adds: order-free and reproducible). The caller solves the rigid fit. */
int tm_icp_accumulate(tm_scene* scene, tm_model* model, const float pose16[16], float dist_thres,
                      uint32_t* count, long long sums15[15]);

/* icp_ (include/impl/scene.hpp:369-400) as one call: starting from pose16, repeat { rigid fit of the current
 * correspondences (Eigen::umeyama without scale, :393) -> correspondences of the fitted pose } while their number
 * does not decrease (:396), at most max_iterations times; every pass, fit and test runs on the device, one
 * read-back at the end. out_pose16 / out_count = the last accepted pose and its number of correspondences
 * (= pose16 and its count when nothing was accepted); the caller extracts the correspondences
 * (tm_extract_correspondences). Same poses as a host loop over tm_icp_accumulate + tm_rigid_fit. */
int tm_icp_refine(tm_scene* scene, tm_model* model, const float pose16[16], float dist_thres,
                  uint32_t max_iterations, float out_pose16[16], uint32_t* out_count, uint32_t* out_iterations);
/* The same for n independent candidates side by side (the candidates of one find_parallel round, which the
 * reference refines one after the other, include/impl/scene.hpp:169-171): poses16 = n x 16 floats; every pass is one
 * launch over (scene blocks, candidates), so the chain of launches is max_iterations + 1 passes long whatever n is.
 * out_poses16 / out_counts / out_iterations (optional) per candidate, each equal to what tm_icp_refine returns for
 * that candidate alone. */
int tm_icp_refine_batch(tm_scene* scene, tm_model* model, const float* poses16, uint32_t n, float dist_thres,
                        uint32_t max_iterations, float* out_poses16, uint32_t* out_counts, uint32_t* out_iterations);
/* The rigid fit itself, on the host: Horn's closed form on the sums of tm_icp_accumulate (count >= 3) */
int tm_rigid_fit(const long long sums15[15], uint32_t count, float pose16[16]);

/* ---- best-hypothesis record and its reduction (new functionality) ---- */
typedef struct tm_best_record {
    uint32_t inlier_count;
    uint32_t hyp_id; /* global hypothesis id, 0xFFFFFFFF = none */
    double score;
    float pose[16];
    uint32_t pad[4];
} tm_best_record; /* 96 bytes */
/* local argmax over a scored batch on the device: order = count desc, score desc, id asc */
int tm_best_of_batch(tm_ctx* ctx, const uint32_t* d_counts, const double* d_scores,
                     const float* d_poses16, uint32_t n_hyp, uint32_t id_offset, uint32_t id_stride,
                     tm_best_record* best);
/* deterministic selection among records (host) */
int tm_best_select(const tm_best_record* records, uint32_t n, tm_best_record* best);

/* ---- multi-GPU: one process per GPU, one RCCL collective per batch (new functionality) ----
 * The reference's only reduction is fut.get() + std::sort over its worker threads
 * (include/impl/scene.hpp:168-195). Here every rank holds a replica of scene / model / table, the sample
 * pairs (bench, multi-model search) or the first-point chunks of find_parallel (scene<Point>) are dealt to
 * the ranks, and the 96-byte records meet in ONE ncclAllGather over xGMI on the context's stream followed
 * by the deterministic selection. RCCL is bound at run time (dlopen librccl.so.1); without it these calls
 * return TM_ERR_UNSUPPORTED. Bootstrap: rank 0 calls tm_comm_get_unique_id and hands the 128 bytes to the
 * other ranks by whatever channel the application has (MPI_Bcast, a file, torch.distributed). */
typedef struct tm_comm tm_comm;
#define TM_COMM_ID_BYTES 128
int tm_comm_get_unique_id(unsigned char id[TM_COMM_ID_BYTES]);
/* collective: every rank calls it with the same id. id may be null for world == 1 (no RCCL state). */
int tm_comm_create(tm_ctx* ctx, int rank, int world, const unsigned char id[TM_COMM_ID_BYTES], tm_comm** comm);
int tm_comm_destroy(tm_comm* comm);
int tm_comm_rank(const tm_comm* comm);
int tm_comm_world(const tm_comm* comm);
/* local argmax with ids that do not depend on the number of ranks: hypothesis id = global sample pair
 * index * query_limit + entry within the pair, global pair = pair_offset + local pair * pair_stride
 * (d_provenance4 as written by tm_generate_hypotheses_dev). The record stays on the device (d_best). */
int tm_best_of_batch_prov(tm_ctx* ctx, const uint32_t* d_counts, const double* d_scores, const float* d_poses16,
                          const uint32_t* d_provenance4, uint32_t n_hyp, uint32_t pair_offset,
                          uint32_t pair_stride, uint32_t query_limit, tm_best_record* d_best);
/* collective: d_local = this rank's n_models records ON THE DEVICE (no host hop); best = the n_models
 * global winners (host), identical on every rank. comm may be null (single process). */
int tm_best_reduce(tm_ctx* ctx, tm_comm* comm, const tm_best_record* d_local, uint32_t n_models,
                   tm_best_record* best);
/* The same in two halves, so that the exchange overlaps the next batch: _begin queues all-gather, selection and
 * the copy into pinned host memory on the context's stream and returns a ticket without waiting (at most 4 in
 * flight per context in any order of ends, TM_ERR_CAPACITY beyond — checked before anything collective, so ranks
 * that make the same sequence of calls fail together); _end waits for that reduction alone and hands out the
 * winners. A ticket is opaque and good once: it names the slot and that use of it, a stale one is TM_ERR_INVALID
 * even after the slot has been handed out again.
 * Counterpart in the reference: fut.get() of the worker futures, include/impl/scene.hpp:168-171. */
int tm_best_reduce_begin(tm_ctx* ctx, tm_comm* comm, const tm_best_record* d_local, uint32_t n_models, int* ticket);
int tm_best_reduce_end(tm_ctx* ctx, int ticket, tm_best_record* best);
/* TM_OK when RCCL can be used by this process (loaded, entry points bound), TM_ERR_UNSUPPORTED otherwise. Not a
 * collective: ranks exchange the answers over their bootstrap channel before any of them calls tm_comm_create. */
int tm_comm_probe(void);
/* collective all-gather of host records (find_parallel's per-chunk candidates): all[r * n + k] */
int tm_comm_allgather_records(tm_comm* comm, const tm_best_record* local, uint32_t n, tm_best_record* all);

/* ---- traits projector math (host side; replaces the statics of the include/<kind>_traits headers) ---- */
int   tm_traits_init_from_samples(int kind, const float* model_state, const float* samples,
                                  float* state); /* returns 1 if the sample set is rejected */
int   tm_traits_project(int kind, const float* state, const float xyz[3], float uvw[3]);
void  tm_traits_unproject(int kind, const float* state, const float uvw[3], float xyz[3]);
void  tm_traits_tangent(int kind, const float* state, const float* surfel, float out[3]);
void  tm_traits_normal(int kind, const float* state, const float* surfel, float out[3]);
float tm_traits_intrinsic_distance(int kind, const float* state, const float p0[3],
                                   const float p1[3]);
/* batched device projection of a cloud into uvw space (pos, traits tangent, traits normal):
 * out surfels have pos = uvw; valid[i] = 0 where project() returns nullopt */
int tm_traits_project_cloud(tm_ctx* ctx, int kind, const float* state, const float* surfels,
                            uint32_t n, float* uvw_surfels, int32_t* valid);

/* The same projection kept on the device: a new scene whose points are the uvw images of `src`'s points
 * under the projector (traits_kind, state) — positions = project(), tangent / normal = traits tangent() /
 * normal(), tangent_mask = |projected tangent| > 0.7 (include/impl/scene.hpp:49-52), mask_ inherited; points
 * whose project() is nullopt never take part. Indices are those of `src`. This is the per-projector step
 * of the traits search (model<Proj,Point> / scene<Proj,Point>, include/impl/cylinder_traits.hpp:92-107,
 * include/impl/plane2_traits.hpp:70-84): one pass over the cloud in HBM, then the identity kernels run
 * on the result with the model's metric (tm_model_set_metric). */
int tm_scene_create_projected(tm_scene* src, int traits_kind, const float* state, tm_scene** out);
/* The two counts a caller rejects a projector candidate by, without building its scene: the points that project
 * and are not masked (= tm_scene_live_count of what tm_scene_create_projected would return) and the tangent
 * points among them with abs(w) <= band (band < 0: all of them). One pass over the resident scene. */
int tm_scene_probe_projected(tm_scene* scene, int traits_kind, const float* state, float band, uint64_t* live,
                             uint32_t* tangent_on_surface);
/* points of the scene that take part in scoring: finite (for a projected scene: project() did not return
 * nullopt) and not masked */
uint64_t tm_scene_live_count(const tm_scene* scene);
/* the surfels of a scene as the kernels see them (uvw surfels for a projected scene) and its tangent_mask */
int tm_scene_download(const tm_scene* scene, float* surfels, int32_t* tangent_mask);
/* The points of a scene with tangent_mask set — ascending indices and their surfels as the kernels see them (uvw
 * surfels for a projected scene): all that the host side of a search ever reads of a scene's cloud (first and second
 * sample points, include/impl/scene.hpp:266-302). A projected scene keeps exactly this on the host (its other points
 * stay on the device; tm_scene_download fetches them when asked). indices / surfels may be null to get *count. */
int tm_scene_tangent_points(const tm_scene* scene, uint32_t* indices, float* surfels, uint32_t cap, uint32_t* count);

/* ---- HIP equivalents of the reference's orphaned OpenCL kernels (opencl/icp.cl, cylinder.cl) ----
 * icp_projection (opencl/icp.cl:1-53, with uv_project opencl/cylinder.cl:1-25 and mat44_multiply
 * opencl/util.cl:1-9): per scene point align -> cylinder unroll -> normalise -> pixel lookup in the
 * model's uv image -> correspondence within max_corr_dist. Outputs exactly as the OpenCL kernel:
 * out_positions4[i] = uv_nrm (or (0,0,0,dist) / zeros), model_indices[i] = pixel index or -1,
 * scene_indices[i] = i or -1. *n_corr = number of correspondences (wave-ballot count). */
int tm_cl_icp_projection(tm_ctx* ctx, const float* pnts4, int n, const float* image4, const int img_size[2],
                         const int img_margin[2], const float* mat_align16, const float* mat_uvw16,
                         const float* mat_proj16, const float* mat_norm16, float max_corr_dist,
                         float* out_positions4, int* model_indices, int* scene_indices, uint32_t* n_corr);
/* icp_correlation (opencl/icp.cl:55-86) fused with the sum the reference never shipped: returns the
 * 3x3 cross-covariance sum_i (scene[is_i]-cs)(model[im_i]-cm)^T / (n-1), row-major in cov9
 * (cov9[3*r+c] = scene component c x model component r, the s0..s8 order of the OpenCL kernel). */
int tm_cl_icp_correlation(tm_ctx* ctx, const float* scene4, uint32_t n_scene, const float* model4,
                          uint32_t n_model, const int* indices_scene, const int* indices_model, int n,
                          const float centroid_scene[4], const float centroid_model[4], double cov9[9]);

#ifdef __cplusplus
}
#endif
#endif /* TM_HIP_H_ */
