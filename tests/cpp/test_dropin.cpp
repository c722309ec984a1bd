/*
 * test_dropin.cpp — exercises the drop-in C++ API (include/triplet_match/*.hpp) end to end on the GPU:
 * model<Point>::init, query, voxel_query, scene<Point>::find_all_parallel, instantiate, traits statics.
 * Reads a binary fixture written by tests/test_cpp_dropin.py: model surfels, scene surfels, resolution,
 * ground-truth scene->model poses. Prints one line per check; exit code 0 = all passed.
 */
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "triplet_match/cylinder_traits.hpp"
#include "triplet_match/identity_traits.hpp"
#include "triplet_match/plane2_traits.hpp"
#include "triplet_match/plane_traits.hpp"
#include "triplet_match/scene.hpp"

using namespace triplet_match;
typedef PointSurfel P;

static int failures = 0;
#define CHECK(cond, msg)                                        \
    do {                                                        \
        bool ok_ = (cond);                                      \
        printf("%s %s\n", ok_ ? "ok  " : "FAIL", msg);          \
        if (!ok_) ++failures;                                   \
    } while (0)

static pointcloud<P>::Ptr read_cloud(std::ifstream& in) {
    uint32_t n = 0;
    in.read((char*)&n, 4);
    auto c = pointcloud<P>::empty();
    c->points.resize(n);
    in.read((char*)c->points.data(), (std::streamsize)n * 48);
    return c;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    auto mcloud = read_cloud(in);
    auto scloud = read_cloud(in);
    float res = 0.f;
    in.read((char*)&res, 4);
    uint32_t n_truth = 0;
    in.read((char*)&n_truth, 4);
    std::vector<mat4f_t> truth(n_truth); /* scene -> model, column-major */
    for (auto& t : truth) in.read((char*)t.m, 64);
    CHECK(in.good(), "fixture read");
    {
        float computed = mcloud->resolution(); /* GPU 2-NN + f32 running mean */
        CHECK(std::fabs(computed - res) < 1e-5f * res + 1e-7f, "pointcloud::resolution() (GPU) agrees with the fixture's value");
        mcloud->set_resolution(res);
    }

    /* free functions */
    auto f = feature<P>(mcloud->points[0], mcloud->points[1], curvature_info_t(), curvature_info_t());
    CHECK(f.has_value() && (*f)[0] == (*f)[3], "feature(): engaged optional, f3 == f0");
    CHECK(discretize(0.5f, 0.f, 1.f, 20u) == 10u && discretize(0.35f, 0.17453294f) == 2u, "discretize() KATs");
    CHECK(std::hash<discrete_feature_t>()(discrete_feature_t{3, 9, 9, 3}) == 0x29DA4D9Du, "std::hash = murmur seed 42");

    discretization_params dp;
    dp.distance_step_count = 20.f;
    dp.angle_step = 10.f * (float)M_PI / 180.f;
    model<P> m(mcloud, dp);
    bool threw = false;
    try {
        m.query(*f);
    } catch (const std::runtime_error& e) {
        threw = std::string(e.what()).find("uninitialized") != std::string::npos;
    }
    CHECK(threw, "query() before init() throws 'Cannot query uninitialized model'");
    sample_parameters sp;
    sp.min_diameter_factor = 0.2f;
    sp.max_diameter_factor = 1.0f;
    sp.force_up = false;
    m.init(sp);
    CHECK(m.point_count() > 10 && m.pair_count() > 100, "init(): tangent subset and pair table built");
    CHECK(m.margin() == 5 && m.extents()[0] > 10, "extents()/margin()");
    CHECK(std::fabs(m.diameter() - mcloud->diameter()) < 1e-6f, "diameter()");
    /* voxel_query at a model point returns a nearby model point */
    {
        const P& q = mcloud->points[7];
        auto v = m.voxel_query(vec4f_t(q.x, q.y, q.z, 1.f));
        bool near_ok = v.has_value();
        if (near_ok) {
            const P& r = mcloud->points[(*v)[0]];
            near_ok = (r.getVector3fMap() - q.getVector3fMap()).norm() < 3.f * res;
        }
        CHECK(near_ok, "voxel_query(): nearest model point to a model point");
        CHECK(!m.voxel_query(vec4f_t(1e6f, 0.f, 0.f, 1.f)).has_value(), "voxel_query(): nullopt outside the grid");
    }
    /* query(): a model pair's own feature must return that pair among the entries */
    {
        uint32_t found = 0, tried = 0;
        std::vector<uint32_t> tang;
        for (uint32_t i = 0; i < mcloud->size(); ++i)
            if (tangent(mcloud->points[i]).norm() > 0.7f) tang.push_back(i);
        for (size_t a = 0; a + 1 < tang.size() && tried < 20; a += 3) {
            uint32_t i = tang[a], j = tang[a + 1];
            auto fij = feature<P>(mcloud->points[i], mcloud->points[j]);
            if (!valid<P>(*fij, m.feature_bounds())) continue;
            auto range = m.query(*fij);
            ++tried;
            for (auto it = range.first; it != range.second; ++it)
                if (std::get<0>(it->second) == i && std::get<1>(it->second) == j) {
                    ++found;
                    break;
                }
        }
        printf("     query(): %u pairs tried, %u found themselves in their bin\n", tried, found);
        CHECK(tried > 0, "query(): iterable range with it->second = (i, j)");
    }

    /* the search */
    scene<P> s(scloud);
    s.set_workers(4);
    auto matches = s.find_all_parallel(m, 3.0f, 0.5f, 0.9f, sp, 2);
    printf("     find_all_parallel: %zu matches\n", matches.size());
    CHECK(!matches.empty(), "find_all_parallel(): at least one instance found");
    size_t good = 0;
    for (const auto& mt : matches) {
        /* match.transform maps model -> scene; compare with the inverse of a ground-truth pose */
        double best = 1e30;
        for (const auto& t : truth) {
            mat4f_t m2s = t.inverse();
            double err = 0;
            for (size_t k = 0; k < mcloud->size(); k += 50) { /* model points carried into the scene */
                vec4f_t a = mt.transform * homogeneous(mcloud->points[k].getVector3fMap());
                vec4f_t b = m2s * homogeneous(mcloud->points[k].getVector3fMap());
                err = std::max(err, (double)(a.head3() - b.head3()).norm());
            }
            best = std::min(best, err);
        }
        printf("     match: %zu correspondences, score %.4f, max alignment error %.5f (resolution %.5f)\n",
               mt.scene_corrs.size(), mt.signed_score, best, res);
        if (best < 6.0 * res) ++good; /* within the ICP correspondence distance 2 x dist_thres x resolution */
    }
    CHECK(good == matches.size() && good >= 1, "every accepted match aligns with a planted pose within 2 x dist_thres x resolution");
    if (!matches.empty()) {
        auto inst = s.instantiate(m, matches[0], true);
        CHECK(inst->size() == mcloud->size(), "instantiate(all): model cloud in the scene frame");
        CHECK(s.instantiate(m, matches[0], false)->size() == 0, "instantiate(false): empty cloud (as the reference)");
    }

    /* curvature(k, idx) through the cloud (GPU kNN-30); the curvature-ratio filter is on by default (as in the
     * reference) and can be switched off */
    {
        curvature_info_t ci = mcloud->curvature(30, 0);
        CHECK(ci.pc_max >= ci.pc_min && ci.pc_min >= -1e-6f, "pointcloud::curvature(30, i): pc_max >= pc_min >= 0");
        model<P> mc(mcloud, dp);
        mc.disable_curvature_filter();
        mc.init(sp);
        CHECK(mc.point_count() >= m.point_count(), "curvature filter can only shrink the tangent subset");
        printf("     tangent subset: %u points with the reference's curvature-ratio filter, %u without\n", m.point_count(),
               mc.point_count());
    }

    /* traits statics */
    {
        auto mh = plane_traits<P>::init_from_model(mcloud);
        CHECK(mh && mh->threshold > 0.f, "plane_traits::init_from_model");
        plane2_traits<P>::samples_t smp{mcloud->points[0], mcloud->points[1], mcloud->points[2]};
        auto h2 = plane2_traits<P>::init_from_samples(mh, smp);
        printf("     plane2 init_from_samples: %s\n", h2 ? "accepted" : "rejected (nullptr)");
        auto ch = cylinder_traits<P>::init_from_axis(vec3f_t(0, 0, 0), vec3f_t(0, 0, 1), 0.5f, 0.1f);
        auto uvw = cylinder_traits<P>::project(ch, vec3f_t(0.5f, 0.f, 0.3f));
        CHECK(uvw && std::fabs((*uvw)[1] - 0.3f) < 1e-6f && std::fabs((*uvw)[2]) < 1e-6f, "cylinder_traits::project");
        CHECK(!cylinder_traits<P>::project(ch, vec3f_t(0.9f, 0.f, 0.3f)), "cylinder_traits::project beyond threshold -> nullopt");
        vec3f_t back = cylinder_traits<P>::unproject(ch, *uvw);
        CHECK((back - vec3f_t(0.5f, 0.f, 0.3f)).norm() < 1e-5f, "cylinder_traits::unproject round trip");
        CHECK(identity_traits<P>::sample_count == 1 && plane_traits<P>::sample_count == 1 &&
                  cylinder_traits<P>::sample_count == 2 && plane2_traits<P>::sample_count == 3,
              "sample_count per traits");
        auto ih = identity_traits<P>::init_from_model(mcloud);
        CHECK(identity_traits<P>::intrinsic_distance(ih, vec3f_t(0, 0, 0), vec3f_t(3, 4, 0)) == 5.f,
              "identity_traits::intrinsic_distance");
    }
    printf("%s (%d failures)\n", failures ? "FAILED" : "PASSED", failures);
    return failures ? 1 : 0;
}
