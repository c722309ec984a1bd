/*
 * test_dropin.cpp — exercises the drop-in C++ API (include/triplet_match/*.hpp) end to end on the GPU:
 * model<Point>::init, query, voxel_query, scene<Point>::find_all_parallel, instantiate, traits statics.
 * Reads a binary fixture written by tests/test_cpp_dropin.py: model surfels, scene surfels, resolution,
 * ground-truth scene->model poses. Prints one line per check; exit code 0 = all passed.
 */
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <fstream>
#include <functional>
#include <thread>

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
        curvature_info_t ci = mcloud->curvature(30u, 0u);
        CHECK(ci.pc_max >= ci.pc_min && ci.pc_min >= -1e-6f, "pointcloud::curvature(30, i): pc_max >= pc_min >= 0");
        model<P> mc(mcloud, dp);
        mc.disable_curvature_filter();
        mc.init(sp);
        CHECK(mc.point_count() >= m.point_count(), "curvature filter can only shrink the tangent subset");
        printf("     tangent subset: %u points with the reference's curvature-ratio filter, %u without\n", m.point_count(),
               mc.point_count());
    }

    /* neighbour accessors of the cloud (include/pointcloud:38-63) against an exhaustive scan with the same f32
     * expression: order (squared distance, index), squared distances, exclusive = inclusive without the first */
    {
        auto brute = [&](const pointcloud<P>& c, const P& q) {
            std::vector<std::pair<float, int>> nb;
            for (size_t j = 0; j < c.size(); ++j) {
                float dx = c.points[j].x - q.x, dy = c.points[j].y - q.y, dz = c.points[j].z - q.z;
                nb.push_back({(dx * dx + dy * dy) + dz * dz, (int)j});
            }
            std::sort(nb.begin(), nb.end());
            return nb;
        };
        bool knn_ok = true, rad_ok = true, ex_ok = true;
        for (uint32_t idx : {0u, 17u, 500u, (uint32_t)mcloud->size() - 1u}) {
            auto ref = brute(*mcloud, mcloud->points[idx]);
            for (uint32_t k : {1u, 7u, 30u, 40u}) { /* 40 > 32: the one-pass route */
                auto nb = mcloud->knn_inclusive(k, idx);
                knn_ok = knn_ok && nb.first.size() == k && nb.second.size() == k;
                for (uint32_t j = 0; j < k && knn_ok; ++j)
                    knn_ok = nb.first[j] == ref[j].second && nb.second[j] == ref[j].first;
                auto ex = mcloud->knn_exclusive(k, idx);
                ex_ok = ex_ok && ex.first.size() == k;
                for (uint32_t j = 0; j < k && ex_ok; ++j) ex_ok = ex.first[j] == ref[j + 1].second;
            }
            const float r = 4.f * res;
            auto rn = mcloud->radius_search_inclusive(r, idx);
            size_t want = 0;
            while (want < ref.size() && ref[want].first <= r * r) ++want;
            rad_ok = rad_ok && rn.first.size() == want;
            for (size_t j = 0; j < want && rad_ok; ++j) rad_ok = rn.first[j] == ref[j].second && rn.second[j] == ref[j].first;
            auto rx = mcloud->radius_search_exclusive(r, idx);
            ex_ok = ex_ok && rx.first.size() + 1 == want && (want < 2 || rx.first[0] == ref[1].second);
        }
        CHECK(knn_ok, "knn_inclusive(k, idx): k nearest, (squared distance, index) order, squared distances");
        CHECK(rad_ok, "radius_search_inclusive(r, idx): exactly the points with d2 <= r*r, ordered");
        CHECK(ex_ok, "knn_exclusive / radius_search_exclusive: the inclusive result without its first entry");
        P off = mcloud->points[3];
        off.x += 0.37f * res;
        auto nbp = mcloud->knn_inclusive(5, off);
        auto refp = brute(*mcloud, off);
        bool pt_ok = nbp.first.size() == 5;
        for (uint32_t j = 0; j < 5 && pt_ok; ++j) pt_ok = nbp.first[j] == refp[j].second;
        CHECK(pt_ok, "knn_inclusive(k, point): a query that is not a cloud point");
        /* curvature(float r, idx) = principal curvatures over radius_search_inclusive(r, idx); with the radius of
         * the 30th neighbour it must equal curvature(30u, idx) whenever no tie straddles that radius */
        auto n30 = mcloud->knn_inclusive(31, 11u);
        if (n30.second[30] > n30.second[29]) {
            const float r30 = std::sqrt(0.5f * (n30.second[29] + n30.second[30]));
            curvature_info_t a = mcloud->curvature(30u, 11u), b = mcloud->curvature(r30, 11u);
            CHECK(a.pc_min == b.pc_min && a.pc_max == b.pc_max, "curvature(float r, idx) == curvature(k, idx) on the same neighbours");
        }
        /* set_indices: the tree holds the subset only; reset */
        auto sub_cloud = pointcloud<P>::empty();
        sub_cloud->points = mcloud->points;
        subset_t evens;
        for (uint32_t i = 0; i < sub_cloud->size(); i += 2) evens.push_back(i);
        sub_cloud->set_indices(evens);
        auto se = sub_cloud->knn_inclusive(6, 1u); /* point 1 itself is not in the tree */
        bool sub_ok = se.first.size() == 6;
        for (int i : se.first) sub_ok = sub_ok && (i % 2 == 0);
        sub_cloud->set_indices(subset_t());
        sub_ok = sub_ok && sub_cloud->knn_inclusive(1, 1u).first[0] == 1;
        CHECK(sub_ok, "set_indices(subset): queries answer from the subset; empty subset = whole cloud again");
    }

    /* query() is re-entrant (the reference's workers call it concurrently, include/impl/scene.hpp:304): ranges
     * of earlier calls stay valid, threads do not disturb each other */
    {
        std::vector<uint32_t> tang;
        for (uint32_t i = 0; i < mcloud->size(); ++i)
            if (tangent(mcloud->points[i]).norm() > 0.7f) tang.push_back(i);
        std::vector<feature_t> fs;
        for (size_t a = 0; a + 1 < tang.size() && fs.size() < 64; a += 2) {
            auto fij = feature<P>(mcloud->points[tang[a]], mcloud->points[tang[a + 1]]);
            if (valid<P>(*fij, m.feature_bounds())) fs.push_back(*fij);
        }
        auto r0 = m.query(fs[0]);
        std::vector<std::pair<uint32_t, uint32_t>> first;
        for (auto it = r0.first; it != r0.second; ++it) first.push_back({std::get<0>(it->second), std::get<1>(it->second)});
        std::vector<uint64_t> sums(4, 0), sums_serial(4, 0);
        auto work = [&](int t, std::vector<uint64_t>& out) {
            uint64_t acc = 0;
            for (int rep = 0; rep < 20; ++rep)
                for (size_t k = t; k < fs.size(); k += 4) {
                    auto r = m.query(fs[k]);
                    for (auto it = r.first; it != r.second; ++it)
                        acc = acc * 1099511628211ull + std::get<0>(it->second) * 65537ull + std::get<1>(it->second);
                }
            out[t] = acc;
        };
        for (int t = 0; t < 4; ++t) work(t, sums_serial);
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t) th.emplace_back(work, t, std::ref(sums));
        for (auto& x : th) x.join();
        bool same = sums == sums_serial;
        size_t k = 0;
        for (auto it = r0.first; it != r0.second; ++it, ++k) /* the first range, after 5,000 other queries */
            same = same && k < first.size() && first[k].first == std::get<0>(it->second) && first[k].second == std::get<1>(it->second);
        CHECK(same && k == first.size() && !first.empty(), "query(): concurrent callers, and a range outlives later queries");
        CHECK((size_t)(r0.second - r0.first) == first.size(), "pair_iter_t: random-access distance");
    }

    /* contexts of the caller's (make_context): three threads, each with its own context, model and scene objects over
     * the same clouds, search at once — the same matches as the search above on the default context, every time; and a
     * scene of one context against a model of another is refused */
    {
        const int T = 3, reps = 4;
        std::vector<context_t> ctxs;
        std::vector<std::unique_ptr<model<P>>> models;
        for (int t = 0; t < T; ++t) {
            ctxs.push_back(make_context(0));
            models.emplace_back(new model<P>(mcloud, dp, ctxs.back()));
            models.back()->init(sp);
        }
        std::vector<int> bad(T, 0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                for (int r = 0; r < reps; ++r) {
                    scene<P> mine(scloud, ctxs[t]); /* a new scene object per repetition: the whole preparation runs */
                    mine.set_workers(4);
                    auto got = mine.find_all_parallel(*models[t], 3.0f, 0.5f, 0.9f, sp, 2);
                    bool same = got.size() == matches.size();
                    for (size_t k = 0; k < got.size() && same; ++k)
                        same = got[k].scene_corrs == matches[k].scene_corrs && got[k].model_corrs == matches[k].model_corrs &&
                               std::memcmp(got[k].transform.m, matches[k].transform.m, 64) == 0;
                    if (!same) ++bad[t];
                }
            });
        for (auto& x : th) x.join();
        CHECK(bad[0] + bad[1] + bad[2] == 0, "three contexts searching at once: every search equals the single-context one");
        bool refused = false;
        try {
            scene<P> other(scloud, ctxs[1]);
            other.find_all_parallel(*models[0], 3.0f, 0.5f, 0.9f, sp, 2);
        } catch (const std::exception&) {
            refused = true;
        }
        CHECK(refused, "a scene of one context against a model of another is refused");
    }

    /* an in-place edit of the scene cloud between two searches is seen (the reference re-reads cloud_ per call):
     * shift every point; the matches must move with them */
    if (!matches.empty()) {
        const vec3f_t shift(0.25f, -0.5f, 0.125f);
        for (auto& p : scloud->points) {
            p.x += shift[0];
            p.y += shift[1];
            p.z += shift[2];
        }
        auto moved = s.find_all_parallel(m, 3.0f, 0.5f, 0.9f, sp, 2);
        bool ok_moved = moved.size() == matches.size();
        for (size_t k = 0; k < moved.size() && ok_moved; ++k) {
            double worst = 1e30;
            for (const auto& old : matches) {
                double e = 0;
                for (size_t q = 0; q < mcloud->size(); q += 97) {
                    vec4f_t a = moved[k].transform * homogeneous(mcloud->points[q].getVector3fMap());
                    vec4f_t b = old.transform * homogeneous(mcloud->points[q].getVector3fMap());
                    e = std::max(e, (double)(a.head3() - (b.head3() + shift)).norm());
                }
                worst = std::min(worst, e);
            }
            ok_moved = worst < 2.0 * res;
        }
        CHECK(ok_moved, "find_all_parallel(): points edited in place between two calls are searched as edited");
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
