/*
 * tm_oracle.h — C interface of the CPU ORACLE (test infrastructure, not product).
 *
 * This library is a plain-C++17 CPU restatement of the triplet-feature RANSAC hot path of
 * richard-vock/triplet_match (reference at /root/reference, file:line cited per function in
 * tm_oracle.cpp). It exists ONLY as the parity checker for the HIP path and as the timed CPU
 * baseline: nothing under triplet_match_amd/ may include, link, import or execute it. Allowed
 * users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures and cannot be built in
 * this image (needs Eigen3, PCL/FLANN, Boost, range-v3). The restatement is pinned only by public
 * MurmurHash3_x86_32 vectors, hand-derived known-answer values and
 * analytic planted-pose scenes. See DESIGN.md.
 *
 * Data layouts:
 *   surfel  : 12 x f32 = 48 B  [x y z 1 | nx ny nz 0 | rgba tx ty tz]   (pcl::PointSurfel; the
 *             tangent aliases data_c[1..3], reference include/common:62-70)
 *   pose    : 16 x f32 column-major 4x4 (Eigen default; opencl/util.cl:2)
 *   feature : 4 x f32, key: 4 x u32, table entry: (u32 i, u32 j)
 */
#ifndef TM_ORACLE_H_
#define TM_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tmo_model tmo_model;
typedef struct tmo_scene tmo_scene;

/* ---- scalar pieces (a2..a5, a10) ---- */
uint32_t tmo_murmur3_32(const uint32_t* words, int n_words, uint32_t seed);
uint32_t tmo_murmur_key(const uint32_t* key, int dim); /* seed 42, reference murmur<Dim> */
uint32_t tmo_discretize_range(float value, float min_value, float range_value, uint32_t steps);
uint32_t tmo_discretize_step(float value, float step_size);
float    tmo_atan2f(float y, float x);
void     tmo_feature(const float* surfel0, const float* surfel1, float f[4]);
void     tmo_discretize_feature(const float f[4], const float bmin[4], const float bmax[4],
                                float distance_step_count, float angle_step, uint32_t key[4]);
int      tmo_valid(const float f[4], const float bmin[4], const float bmax[4]);
void     tmo_valid_bounds(const float bmin[4], const float bmax[4], float min_rel_dist,
                          float max_rel_dist, float out_min[4], float out_max[4]);
void     tmo_base_transform(const float src_i[3], const float src_j[3], const float src_t[3],
                            const float tgt_i[3], const float tgt_j[3], const float tgt_t[3],
                            float pose16[16]);
uint32_t tmo_early_drop_upper(uint32_t tried, uint32_t subset_size, uint32_t corrs);
/* literal transcription of scene.hpp:493-498 (relies on x86-64 wrap of negative double->u32) */
uint32_t tmo_early_drop_upper_literal(uint32_t tried, uint32_t subset_size, uint32_t corrs);
void     tmo_early_drop_checkpoints(uint32_t subset_size, uint32_t tests[18]);

/* how many angle bins change when glibc's atan2f (the reference's call, feature.hpp:7) replaces tmo_atan2f: over the
 * model's filtered subset pairs (ij == NULL) or an explicit pair list; see tm_oracle.cpp */
uint64_t tmo_atan2_census(const float* surfels, uint32_t n, const uint32_t* subset, uint32_t n_subset, const uint32_t* ij,
                          uint64_t n_ij, float lower_bound, float upper_bound, float angle_step, uint32_t* differing_ij,
                          uint32_t cap, uint64_t* n_pairs, uint64_t* n_bits, float* max_abs);
/* restated early-drop bound against the literal transcription over a grid; returns the number of differences */
uint64_t tmo_early_drop_compare(uint32_t subset_max, uint32_t step_subset, uint32_t step_tried, uint32_t step_corrs,
                                uint64_t* n_points, uint32_t* bad3, uint32_t cap);

/* cloud->resolution() (pointcloud.hpp:66-82): exhaustive 2-NN + f32 running mean in point order */
float    tmo_cloud_resolution(const float* surfels, uint32_t n, float* nn_dist, int threads);

/* cloud->curvature(k, i) for all i (pointcloud.hpp:3-44,200-204): curv5 = (dir xyz, pc_min, pc_max) */
void     tmo_cloud_curvature(const float* surfels, uint32_t n, uint32_t k, float* curv5, int threads);

/* knn_inclusive(k, point) / radius_search_inclusive(r, point) (pointcloud.hpp:138-146,168-176), exhaustive,
 * order (squared distance, index); subset = set_indices (NULL, 0 = the whole cloud). Return the number found. */
uint32_t tmo_cloud_knn(const float* surfels, uint32_t n, const uint32_t* subset, uint32_t n_subset,
                       const float query[3], uint32_t k, int32_t* indices, float* sq_dists);
uint32_t tmo_cloud_radius_search(const float* surfels, uint32_t n, const uint32_t* subset, uint32_t n_subset,
                                 const float query[3], float radius, int32_t* indices, float* sq_dists,
                                 uint32_t cap);
/* principal_curvatures over an explicit neighbour list (pointcloud.hpp:3-44; curvature(float r, idx) :206-209) */
void     tmo_cloud_curvature_of(const float* surfels, uint32_t p_idx, const int32_t* indices, uint32_t n_idx,
                                float curv5[5]);

/* ---- model (a6..a9) ---- */
tmo_model* tmo_model_create(const float* surfels, uint32_t n, float resolution,
                            float distance_step_count, float angle_step);
void     tmo_model_destroy(tmo_model* m);
/* voxel NN grid; accelerated exact search (bucket grid + ring expansion). returns 0 ok */
/* include/impl/model.hpp:17-38: bbox / diameter / voxel grid over the finite points of init()'s subset */
int      tmo_model_set_grid_subset(tmo_model* m, const uint32_t* subset, uint32_t n);
int      tmo_model_build_voxels(tmo_model* m, int threads);
/* same result by exhaustive O(V*M) scan — for cross-checking the accelerated build on tiny inputs */
int      tmo_model_build_voxels_bruteforce(tmo_model* m, int threads);
/* import a voxel table computed elsewhere (bench cpu_baseline at sizes where parity is already shown) */
int      tmo_model_set_voxels(tmo_model* m, const uint32_t* voxels, uint64_t count);
int      tmo_model_build_pairs(tmo_model* m, const uint32_t* subset, uint32_t n_subset,
                               float min_diameter_factor, float max_diameter_factor);
float    tmo_model_diameter(const tmo_model* m);
/* the model is a projector's uvw cloud: distances use that traits' intrinsic_distance */
void     tmo_model_set_metric(tmo_model* m, int traits_kind, float radius);
uint32_t tmo_model_size(const tmo_model* m);
uint32_t tmo_model_point_count(const tmo_model* m); /* tangent-subset size */
void     tmo_model_get_grid(const tmo_model* m, int extents[3], float to_voxel16[16]);
uint64_t tmo_model_voxel_count(const tmo_model* m);
const uint32_t* tmo_model_voxels(const tmo_model* m);
void     tmo_model_get_feature_bounds(const tmo_model* m, float bmin[4], float bmax[4]);
uint64_t tmo_model_pair_count(const tmo_model* m);     /* entries in the table */
uint64_t tmo_model_valid_count(const tmo_model* m);    /* pass-1 valid pairs */
/* returns -1 if not initialised (reference throws), else 0; entries in container order */
int      tmo_model_query(const tmo_model* m, const float f[4], uint32_t* pairs_ij, uint32_t cap,
                         uint32_t* count_total);
int      tmo_model_query_key(const tmo_model* m, const uint32_t key[4], uint32_t* pairs_ij,
                             uint32_t cap, uint32_t* count_total);
/* dump table: distinct keys (4 u32 each) and counts, sorted by key lexicographically */
uint32_t tmo_model_table_keys(const tmo_model* m, uint32_t* keys4, uint32_t* counts, uint32_t cap);
/* returns 1 and index if inside grid, 0 if nullopt */
int      tmo_model_voxel_query(const tmo_model* m, const float pos4[4], uint32_t* index);

/* ---- scene (a10..a14) ---- */
tmo_scene* tmo_scene_create(const float* surfels, uint32_t n, const int32_t* tangent_mask);
void     tmo_scene_destroy(tmo_scene* s);
void     tmo_scene_set_mask(tmo_scene* s, const int32_t* mask); /* NULL clears */

/* project_ (scene.hpp:411-510). subset NULL => all points 0..n-1 (finish_find, :100-106).
 * corr buffers may be NULL. returns number of correspondences. */
uint32_t tmo_project(const tmo_scene* s, const tmo_model* m, const int32_t* subset,
                     uint32_t n_subset, const float pose16[16], float accept_prob, float dist_thres,
                     int early_out, uint32_t* scene_corrs, uint32_t* model_corrs, double* score,
                     uint32_t* saved);

/* inner body of find_in_subset (scene.hpp:288-319) over an explicit list of scene (i,j) pairs.
 * Output compacted in visit order: pose (16 f32) + provenance (pair_idx, m_i, m_j, entry_rank). */
int      tmo_generate_hypotheses(const tmo_scene* s, const tmo_model* m, const uint32_t* ij_pairs,
                                 uint32_t n_pairs, float min_diameter_factor,
                                 float max_diameter_factor, int force_up, int query_limit,
                                 float* poses16, uint32_t* provenance4, uint32_t cap,
                                 uint32_t* count);

/* multi-threaded driver used for the timed CPU baseline and batch parity: one project_ per pose */
void     tmo_score_hypotheses(const tmo_scene* s, const tmo_model* m, const float* poses16,
                              uint32_t n_hyp, const int32_t* subset, uint32_t n_subset,
                              float accept_prob, float dist_thres, int early_drop, int threads,
                              uint32_t* counts, double* scores, uint64_t* points_evaluated);

/* radius_search_inclusive(diameter, p1) membership, ascending index order (impl/scene.hpp:273) */
uint32_t tmo_scene_radius_subset(const tmo_scene* s, uint32_t center_idx, float radius,
                                 int32_t* out, uint32_t cap);

/* ---- traits math (a15): state = [radius, threshold, origin(3), g2l(16), l2g(16)] = 37 f32 ---- */
#define TMO_TRAITS_IDENTITY 0
#define TMO_TRAITS_PLANE 1
#define TMO_TRAITS_PLANE2 2
#define TMO_TRAITS_CYLINDER 3
#define TMO_STATE_FLOATS 37
/* samples: sample_count surfels (identity 1, plane 1, cylinder 2, plane2 3). returns 0, or 1 if the
 * sample set is rejected (reference returns nullptr, plane2_traits.hpp:66-68) */
int      tmo_traits_init_from_samples(int kind, const float* model_state, const float* samples,
                                      float* state);
/* returns 1 if projected, 0 if nullopt */
int      tmo_traits_project(int kind, const float* state, const float xyz[3], float uvw[3]);
void     tmo_traits_unproject(int kind, const float* state, const float uvw[3], float xyz[3]);
void     tmo_traits_tangent(int kind, const float* state, const float* surfel, float out[3]);
void     tmo_traits_normal(int kind, const float* state, const float* surfel, float out[3]);
float    tmo_traits_intrinsic_distance(int kind, const float* state, const float p0[3],
                                       const float p1[3]);
/* opencl/cylinder.cl:1-25 uv_project and opencl/icp.cl:1-53 icp_projection for one point */
void     tmo_cl_uv_project(const float loc4[4], const float cyl2ncoord16[16], float out4[4]);
void     tmo_cl_icp_projection(const float* pnts4, int n, const float* image4, const int img_size[2],
                               const int img_margin[2], const float* mat_align, const float* mat_uvw,
                               const float* mat_proj, const float* mat_norm, float max_corr_dist,
                               float* out_positions4, int* model_indices, int* scene_indices);

#ifdef __cplusplus
}
#endif
#endif /* TM_ORACLE_H_ */
