/*
 * test_refnames.cpp — a caller written against the reference's installed headers compiles unchanged: the
 * reference installs include/ as include/triplet_match/ with extension-less names (CMakeLists.txt:18) and its
 * callers write `#include <scene>` with that directory on the include path (apps/triplet_match.cpp:7) or
 * `#include <triplet_match/scene>`. Compiled both ways by tests/test_cpp_dropin.py (compile + link only; no GPU).
 */
#ifdef TM_TEST_BARE_NAMES
#include <cartesian_tuple>
#include <common>
#include <cylinder_traits>
#include <discretize>
#include <feature>
#include <identity_traits>
#include <model>
#include <plane2_traits>
#include <plane_traits>
#include <pointcloud>
#include <scene>
#else
#include <triplet_match/cartesian_tuple>
#include <triplet_match/common>
#include <triplet_match/cylinder_traits>
#include <triplet_match/discretize>
#include <triplet_match/feature>
#include <triplet_match/identity_traits>
#include <triplet_match/model>
#include <triplet_match/plane2_traits>
#include <triplet_match/plane_traits>
#include <triplet_match/pointcloud>
#include <triplet_match/scene>
#endif

#include <cstdio>
#include <type_traits>

using namespace triplet_match;
typedef PointSurfel P;

int main() {
    /* the typedef block of include/model:17-28 and include/scene:12-21 */
    static_assert(std::is_same<model<P>::wptr_t, std::weak_ptr<model<P>>>::value, "model::wptr_t");
    static_assert(std::is_same<model<P>::const_wptr_t, std::weak_ptr<const model<P>>>::value, "model::const_wptr_t");
    static_assert(std::is_same<scene<P>::wptr_t, std::weak_ptr<scene<P>>>::value, "scene::wptr_t");
    static_assert(std::is_same<scene<P>::const_wptr_t, std::weak_ptr<const scene<P>>>::value, "scene::const_wptr_t");
    static_assert(std::is_same<model<P>::result_t, std::tuple<uint32_t, uint32_t>>::value, "model::result_t");
    static_assert(std::is_same<cylinder_traits<P>::samples_t, repeat_t<P, 2>>::value, "samples_t = repeat_t<Point, N>");
    /* the member functions a caller of the reference uses exist with the reference's shapes (not called here) */
    typedef std::pair<std::vector<int>, std::vector<float>> nb_t;
    nb_t (pointcloud<P>::*knn_i)(uint32_t, uint32_t) const = &pointcloud<P>::knn_inclusive;
    nb_t (pointcloud<P>::*knn_p)(uint32_t, const P&) const = &pointcloud<P>::knn_exclusive;
    nb_t (pointcloud<P>::*rad_i)(float, uint32_t) const = &pointcloud<P>::radius_search_inclusive;
    nb_t (pointcloud<P>::*rad_p)(float, const P&) const = &pointcloud<P>::radius_search_exclusive;
    curvature_info_t (pointcloud<P>::*curv_k)(uint32_t, uint32_t) const = &pointcloud<P>::curvature;
    curvature_info_t (pointcloud<P>::*curv_r)(float, uint32_t) const = &pointcloud<P>::curvature;
    void (pointcloud<P>::*set_idx)(subset_t) = &pointcloud<P>::set_indices;
    void (pointcloud<P>::*reset)() = &pointcloud<P>::reset_tree;
    std::pair<model<P>::pair_iter_t, model<P>::pair_iter_t> (model<P>::*query)(const feature_t&, bool) = &model<P>::query;
    std::printf("%d\n", (knn_i && knn_p && rad_i && rad_p && curv_k && curv_r && set_idx && reset && query) ? 1 : 0);
    std::printf("PASSED\n");
    return 0;
}
