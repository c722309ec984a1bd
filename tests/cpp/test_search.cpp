/*
 * test_search.cpp — runs the host orchestration (include/triplet_match/search.hpp) over the CPU oracle
 * backend (mode "oracle": host-logic test, no GPU) and, in mode "both", also over the HIP backend and
 * requires identical results (same candidates, same correspondences, bit-identical transforms).
 * TEST CODE: the only place that links the oracle next to the product library.
 */
#include <cstdio>
#include <fstream>
#include <thread>
#include <unistd.h>

#include "../../oracle/tm_oracle.h"
#include "triplet_match/scene.hpp"

using namespace triplet_match;
typedef PointSurfel P;

struct oracle_backend {
    tmo_scene* s;
    tmo_model* m;
    const pointcloud<P>* scloud;
    const pointcloud<P>* mcloud;
    uint32_t n_t;
    search::params_t p;
    std::vector<int32_t> mask, tmask;

    uint32_t scene_size() const { return (uint32_t)scloud->size(); }
    uint32_t model_size() const { return (uint32_t)mcloud->size(); }
    uint32_t model_point_count() const { return n_t; }
    float model_diameter() const { return tmo_model_diameter(m); }
    const float* scene_surfel(uint32_t i) const { return scloud->points[i].raw(); }
    const float* model_surfel(uint32_t i) const { return mcloud->points[i].raw(); }
    bool pair_is_valid_sample(uint32_t i, uint32_t j) const {
        /* one pair through the oracle's generator path: valid <=> it survives the filters and valid();
         * query_limit 0 entries would hide validity, so restate the filter with the oracle's scalars */
        if (i == j || mask[i] || !tmask[j] || mask[j]) return false;
        const float* a = scloud->points[i].raw();
        const float* b = scloud->points[j].raw();
        float lower = tmo_model_diameter(m) * p.min_diameter_factor, upper = tmo_model_diameter(m) * p.max_diameter_factor;
        lower *= lower;
        upper *= upper;
        float d[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
        float sq = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
        if (sq > 0.f) {
            float n = sqrtf(sq);
            d[0] /= n, d[1] /= n, d[2] /= n;
        }
        if (sq < lower || sq > upper) return false;
        if (1.f - fabsf((d[0] * a[9] + d[1] * a[10]) + d[2] * a[11]) < 0.01f) return false;
        float f[4], bmin[4], bmax[4];
        tmo_feature(a, b, f);
        tmo_model_get_feature_bounds(m, bmin, bmax);
        return tmo_valid(f, bmin, bmax) != 0;
    }
    bool supports_batched() const { return false; } /* the oracle always takes the one-at-a-time loop */
    void generate_with_pairs(const std::vector<uint32_t>&, std::vector<search::pose_t>&, std::vector<uint32_t>&) {}
    void score_ball(const std::vector<search::pose_t>&, const std::vector<uint32_t>&, float, bool, std::vector<uint32_t>&,
                    std::vector<double>&) {}
    void announce_firsts(const std::vector<uint32_t>&) {}
    void radius_subset(uint32_t i, float radius, std::vector<int32_t>& out) {
        uint32_t n = tmo_scene_radius_subset(s, i, radius, nullptr, 0);
        out.resize(n);
        tmo_scene_radius_subset(s, i, radius, out.data(), n);
    }
    void generate(const std::vector<uint32_t>& ij, std::vector<search::pose_t>& poses) {
        uint32_t n_pairs = (uint32_t)(ij.size() / 2), count = 0;
        tmo_generate_hypotheses(s, m, ij.data(), n_pairs, p.min_diameter_factor, p.max_diameter_factor,
                                p.force_up ? 1 : 0, search::query_limit, nullptr, nullptr, 0, &count);
        poses.resize(count);
        if (count)
            tmo_generate_hypotheses(s, m, ij.data(), n_pairs, p.min_diameter_factor, p.max_diameter_factor,
                                    p.force_up ? 1 : 0, search::query_limit, &poses[0].m[0], nullptr, count, &count);
    }
    void score(const std::vector<search::pose_t>& poses, const std::vector<int32_t>& subset, bool early_drop,
               std::vector<uint32_t>& counts, std::vector<double>& scores) {
        counts.resize(poses.size());
        scores.resize(poses.size());
        tmo_score_hypotheses(s, m, &poses[0].m[0], (uint32_t)poses.size(), subset.data(), (uint32_t)subset.size(),
                             p.accept_prob, p.dist_thres, early_drop ? 1 : 0, 4, counts.data(), scores.data(), nullptr);
    }
    void extract(const search::pose_t& T, float dist_thres, search::candidate_t& out) {
        out.transform = T;
        out.scene_corrs.resize(scloud->size());
        out.model_corrs.resize(scloud->size());
        uint32_t saved = 0;
        uint32_t n = tmo_project(s, m, nullptr, 0, T.m, p.accept_prob, dist_thres, 0, out.scene_corrs.data(),
                                 out.model_corrs.data(), &out.signed_score, &saved);
        out.scene_corrs.resize(n);
        out.model_corrs.resize(n);
    }
    search::detail::icp_sums_t icp_sums(const search::pose_t& T, float dist_thres) {
        search::candidate_t c;
        extract(T, dist_thres, c);
        search::detail::icp_sums_t r;
        for (size_t k = 0; k < c.scene_corrs.size(); ++k)
            search::detail::icp_sums_add(r, scloud->points[c.scene_corrs[k]].raw(), mcloud->points[c.model_corrs[k]].raw());
        return r;
    }
    void set_mask(const std::vector<int32_t>& mk) {
        mask = mk;
        tmo_scene_set_mask(s, mk.data());
    }
};

/* The same backend handing the sampling loop its SPARSE view of a neighbourhood (size + (rank, index) of the tangent
 * members, what tm_scene_radius_tangent_lists gives the HIP backend) instead of the whole list: the search must not
 * notice — same draws, same samples, same candidates (mode "oracle_sparse"). */
struct oracle_backend_sparse : oracle_backend {
    std::vector<int32_t> dense;
    std::vector<uint32_t> pairs;
    bool supports_tangent_view() const { return true; }
    void tangent_view(uint32_t i, float radius, size_t& n_nn, const uint32_t*& rank_index_pairs, size_t& n_pairs) {
        radius_subset(i, radius, dense);
        pairs.clear();
        for (size_t r = 0; r < dense.size(); ++r)
            if (tmask[dense[r]]) {
                pairs.push_back((uint32_t)r);
                pairs.push_back((uint32_t)dense[r]);
            }
        n_nn = dense.size();
        rank_index_pairs = pairs.data();
        n_pairs = pairs.size() / 2;
    }
};

/* Exchange over a shared directory, for the multi-process CPU test of the rank-sharded search (the product
 * exchange is RCCL through tm_comm_allgather_records): every all-gather writes <dir>/x<seq>_r<rank>.bin
 * (atomic rename) and waits for the files of the other ranks. */
struct file_exchange {
    std::string dir;
    int rank_, world_;
    uint32_t seq = 0;
    int rank() const { return rank_; }
    int world() const { return world_; }
    void allgather(const search::chunk_record* local, uint32_t n, search::chunk_record* all) {
        char name[512], tmp[512];
        snprintf(name, sizeof name, "%s/x%u_r%d.bin", dir.c_str(), seq, rank_);
        snprintf(tmp, sizeof tmp, "%s.tmp", name);
        {
            std::ofstream o(tmp, std::ios::binary);
            o.write((const char*)&n, 4);
            o.write((const char*)local, (std::streamsize)n * sizeof(search::chunk_record));
        }
        rename(tmp, name);
        for (int r = 0; r < world_; ++r) {
            snprintf(name, sizeof name, "%s/x%u_r%d.bin", dir.c_str(), seq, r);
            for (int tries = 0;; ++tries) {
                std::ifstream in(name, std::ios::binary);
                uint32_t m = 0;
                if (in.good() && in.read((char*)&m, 4) && m == n &&
                    in.read((char*)(all + (size_t)r * n), (std::streamsize)n * sizeof(search::chunk_record)))
                    break;
                if (tries > 60000) {
                    fprintf(stderr, "file_exchange: rank %d never wrote %s\n", r, name);
                    exit(5);
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        ++seq;
    }
};

static pointcloud<P>::Ptr read_cloud(std::ifstream& in) {
    uint32_t n = 0;
    in.read((char*)&n, 4);
    auto c = pointcloud<P>::empty();
    c->points.resize(n);
    in.read((char*)c->points.data(), (std::streamsize)n * 48);
    return c;
}

/* the search's trace as JSON, in the layout of tests/golden/search_trace_*.json.gz (oracle/ref_search.py) */
static unsigned long long fnv_u32(const std::vector<uint32_t>& v) {
    unsigned long long h = 1469598103934665603ull;
    const unsigned char* d = (const unsigned char*)v.data();
    for (size_t k = 0; k < v.size() * 4; ++k) h = (h ^ d[k]) * 1099511628211ull;
    return h;
}
static void write_pose(FILE* f, const search::pose_t& T) {
    fputc('[', f);
    for (int k = 0; k < 16; ++k) {
        uint32_t b;
        memcpy(&b, &T.m[k], 4);
        fprintf(f, "%s%u", k ? "," : "", b);
    }
    fputc(']', f);
}
static void write_matches(FILE* f, const char* name, const std::vector<search::trace_match>& v) {
    fprintf(f, "\"%s\":[", name);
    for (size_t k = 0; k < v.size(); ++k) {
        fprintf(f, "%s{\"pose\":", k ? "," : "");
        write_pose(f, v[k].transform);
        fprintf(f, ",\"n\":%zu,\"scene\":\"%016llx\",\"model\":\"%016llx\",\"score\":%.17g}", v[k].scene_corrs.size(),
                fnv_u32(v[k].scene_corrs), fnv_u32(v[k].model_corrs), v[k].signed_score);
    }
    fputc(']', f);
}
static void write_trace(const char* path, const search::trace_t& t) {
    FILE* f = fopen(path, "w");
    if (!f) exit(6);
    fprintf(f, "{\"rounds\":[");
    for (size_t r = 0; r < t.rounds.size(); ++r) {
        const auto& R = t.rounds[r];
        fprintf(f, "%s{\"left\":%llu,\"min_points\":%u,\"min_points_all\":%u,\"outer_bound\":%llu,\"batch_size\":%llu,\"outer\":[",
                r ? "," : "", (unsigned long long)R.left, R.min_points, R.min_points_all, (unsigned long long)R.outer_bound,
                (unsigned long long)R.batch_size);
        for (size_t k = 0; k < R.outer.size(); ++k) fprintf(f, "%s%u", k ? "," : "", R.outer[k]);
        fprintf(f, "],\"chunks\":[");
        for (size_t c = 0; c < R.chunks.size(); ++c) {
            const auto& C = R.chunks[c];
            fprintf(f, "%s{\"cons\":%u,\"firsts\":[", c ? "," : "", C.cons);
            for (size_t k = 0; k < C.firsts.size(); ++k) {
                const auto& F = C.firsts[k];
                fprintf(f, "%s{\"i\":%u,\"nn\":%u,\"inner_bound\":%llu,\"draws\":%llu,\"sampled\":%s,\"js\":[", k ? "," : "", F.i,
                        F.nn, (unsigned long long)F.inner_bound, (unsigned long long)F.draws, F.sampled ? "true" : "false");
                for (size_t e = 0; e < F.js.size(); ++e) fprintf(f, "%s%u", e ? "," : "", F.js[e]);
                fprintf(f, "]}");
            }
            fprintf(f, "],\"hit\":");
            if (C.hit) {
                fprintf(f, "{\"pose\":");
                write_pose(f, C.hit_pose);
                fputc('}', f);
            } else
                fprintf(f, "null");
            fputc('}', f);
        }
        fprintf(f, "],");
        write_matches(f, "candidates", R.candidates);
        fputc(',', f);
        write_matches(f, "accepted", R.accepted);
        fputc('}', f);
    }
    fprintf(f, "]}\n");
    fclose(f);
}

static void dump(const char* tag, const std::vector<search::candidate_t>& r) {
    printf("%s: %zu candidates\n", tag, r.size());
    for (const auto& c : r)
        printf("   corrs %zu score %.9f t = (%.6f %.6f %.6f)\n", c.scene_corrs.size(), c.signed_score, c.transform.m[12],
               c.transform.m[13], c.transform.m[14]);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const bool both = std::string(argv[2]) == "both" || std::string(argv[2]) == "both_noearlydrop";
    /* "sharded <dir> <rank> <world>": this process is one rank of a rank-sharded search over the oracle */
    const bool sharded = std::string(argv[2]) == "sharded" && argc >= 6;
    /* "trace_oracle|trace_hip <out.json> <max_icp_iterations> <early_drop 0|1> <workers>": run one search over that backend
     * and write its trace (tests/test_search_traces.py compares it with the committed traces of the Python restatement) */
    const bool trace_oracle = std::string(argv[2]) == "trace_oracle" && argc >= 7;
    const bool trace_hip = std::string(argv[2]) == "trace_hip" && argc >= 7;
    bool no_early_drop = std::string(argv[2]) == "both_noearlydrop" || std::string(argv[2]) == "oracle_noearlydrop";
    std::ifstream in(argv[1], std::ios::binary);
    auto mcloud = read_cloud(in);
    auto scloud = read_cloud(in);
    float res = 0.f;
    in.read((char*)&res, 4);
    uint32_t n_truth = 0;
    in.read((char*)&n_truth, 4);
    std::vector<mat4f_t> truth(n_truth);
    for (auto& t : truth) in.read((char*)t.m, 64);
    if (!in.good()) return 3;

    const float angle_step = 10.f * (float)M_PI / 180.f;
    /* the reference's masks (include/impl/model.hpp:96-99, include/impl/scene.hpp:49-52): |tangent| > 0.7
     * and pc_min / pc_max < 0.2 on curvature(30, i); the curvature comes from the oracle here */
    std::vector<float> mcurv(5 * mcloud->size()), scurv(5 * scloud->size());
    tmo_cloud_curvature(mcloud->points[0].raw(), (uint32_t)mcloud->size(), 30, mcurv.data(), 8);
    tmo_cloud_curvature(scloud->points[0].raw(), (uint32_t)scloud->size(), 30, scurv.data(), 8);
    std::vector<uint32_t> subset;
    for (uint32_t i = 0; i < mcloud->size(); ++i)
        if (tangent(mcloud->points[i]).norm() > 0.7f && (mcurv[5 * (size_t)i + 3] / mcurv[5 * (size_t)i + 4]) < 0.2f)
            subset.push_back(i);
    std::vector<int32_t> tmask(scloud->size());
    uint32_t n_tmask = 0;
    for (uint32_t i = 0; i < scloud->size(); ++i) {
        tmask[i] = (tangent(scloud->points[i]).norm() > 0.7f &&
                    (scurv[5 * (size_t)i + 3] / scurv[5 * (size_t)i + 4]) < 0.2f) ? 1 : 0;
        n_tmask += (uint32_t)tmask[i];
    }
    printf("masks (tangent and curvature ratio): model subset %zu, scene tangent_mask %u\n", subset.size(), n_tmask);

    search::params_t p;
    p.dist_thres = 3.f;
    p.accept_prob = 0.5f;
    p.max_icp_iterations = 2;
    p.workers = 4;
    if (trace_oracle || trace_hip) {
        p.max_icp_iterations = (uint32_t)atoi(argv[4]);
        no_early_drop = atoi(argv[5]) == 0;
        p.workers = (uint32_t)atoi(argv[6]);
    }
    p.early_drop = !no_early_drop;
    if (trace_hip) { /* the product alone: model<P>::init and the scene on the device, search.hpp over the HIP backend */
        discretization_params dp;
        dp.distance_step_count = 20.f;
        dp.angle_step = angle_step;
        mcloud->set_resolution(res);
        model<P> gm(mcloud, dp);
        sample_parameters sp;
        gm.init(sp);
        if (gm.point_count() != subset.size()) {
            printf("FAIL model<P>::init subset (%u) differs from the oracle's masks (%zu)\n", gm.point_count(), subset.size());
            return 1;
        }
        tm_scene* gs = nullptr;
        detail::check(tm_scene_create(gm.context(), scloud->points[0].raw(), (uint32_t)scloud->size(), tmask.data(), &gs),
                      "tm_scene_create");
        detail::hip_backend<P> hb;
        hb.s = gs;
        hb.m = gm.handle();
        hb.scloud = scloud.get();
        hb.mcloud = mcloud.get();
        hb.n_t = gm.point_count();
        hb.diam = gm.diameter();
        hb.p = p;
        search::trace_t tr;
        p.trace = &tr;
        auto rg = search::find_all(hb, tmask, p);
        dump("hip backend", rg);
        write_trace(argv[3], tr);
        tm_scene_destroy(gs);
        printf("PASSED (trace written)\n");
        return 0;
    }

    oracle_backend ob;
    ob.m = tmo_model_create(mcloud->points[0].raw(), (uint32_t)mcloud->size(), res, 20.f, angle_step);
    if (tmo_model_build_voxels(ob.m, 8) || tmo_model_build_pairs(ob.m, subset.data(), (uint32_t)subset.size(), 0.2f, 1.0f))
        return 4;
    ob.s = tmo_scene_create(scloud->points[0].raw(), (uint32_t)scloud->size(), tmask.data());
    ob.scloud = scloud.get();
    ob.mcloud = mcloud.get();
    ob.n_t = (uint32_t)subset.size();
    ob.p = p;
    ob.tmask = tmask;
    std::vector<search::candidate_t> ro;
    if (trace_oracle) {
        search::trace_t tr;
        p.trace = &tr;
        ro = search::find_all(ob, tmask, p);
        dump("oracle backend", ro);
        write_trace(argv[3], tr);
        tmo_scene_destroy(ob.s);
        tmo_model_destroy(ob.m);
        printf("PASSED (trace written)\n");
        return 0;
    }
    if (sharded) {
        file_exchange fx{argv[3], atoi(argv[4]), atoi(argv[5])};
        ro = search::find_all(ob, tmask, p, fx);
    } else {
        ro = search::find_all(ob, tmask, p);
    }
    dump("oracle backend", ro);
    for (const auto& c : ro) { /* fingerprint of the whole result, for the sharded == single-process test */
        unsigned long long h = 1469598103934665603ull;
        auto mix = [&](const void* d, size_t n) {
            for (size_t k = 0; k < n; ++k) h = (h ^ ((const unsigned char*)d)[k]) * 1099511628211ull;
        };
        mix(c.transform.m, 64);
        mix(c.scene_corrs.data(), c.scene_corrs.size() * 4);
        mix(c.model_corrs.data(), c.model_corrs.size() * 4);
        printf("   fingerprint %016llx\n", h);
    }
    int failures = 0;
    if (ro.empty()) {
        printf("FAIL no candidate found by the host logic over the oracle\n");
        ++failures;
    }
    if (std::string(argv[2]) == "oracle_sparse") { /* sparse neighbourhood views: the identical search */
        oracle_backend_sparse sb;
        static_cast<oracle_backend&>(sb) = ob;
        search::stats_t st_sparse, st_dense;
        search::params_t ps = p;
        ps.stats = &st_sparse;
        auto rs = search::find_all(sb, tmask, ps);
        ps.stats = &st_dense;
        auto rd = search::find_all(ob, tmask, ps);
        bool same = rs.size() == ro.size() && rd.size() == ro.size();
        for (size_t k = 0; k < rs.size() && same; ++k)
            same = memcmp(rs[k].transform.m, ro[k].transform.m, 64) == 0 && rs[k].scene_corrs == ro[k].scene_corrs &&
                   rs[k].model_corrs == ro[k].model_corrs;
        same = same && st_sparse.draws == st_dense.draws && st_sparse.firsts_sampled == st_dense.firsts_sampled;
        printf("%s sparse neighbourhood views == whole lists: %zu candidates, %llu draws over %llu first points, %llu entries "
               "handed over instead of %llu\n", same ? "ok  " : "FAIL", rs.size(), (unsigned long long)st_sparse.draws,
               (unsigned long long)st_sparse.firsts_sampled, (unsigned long long)st_sparse.neighbours,
               (unsigned long long)st_dense.neighbours);
        if (!same || st_sparse.draws == 0) ++failures;
    }
    for (const auto& c : ro) { /* each accepted candidate must sit on a planted pose */
        double best = 1e30;
        for (const auto& t : truth) {
            double err = 0;
            for (size_t k = 0; k < c.scene_corrs.size(); k += 7) { /* over the matched instance's points */
                mat4f_t T;
                memcpy(T.m, c.transform.m, 64);
                const auto& q = scloud->points[c.scene_corrs[k]];
                vec4f_t a = T * homogeneous(q.getVector3fMap());
                vec4f_t b = t * homogeneous(q.getVector3fMap());
                err = std::max(err, (double)(a.head3() - b.head3()).norm());
            }
            best = std::min(best, err);
        }
        printf("   alignment error vs nearest planted pose: %.6f (resolution %.6f)\n", best, res);
        if (!(best < 6.0 * res)) { /* within the ICP correspondence distance 2 x dist_thres x resolution */
            printf("FAIL candidate is not a planted pose\n");
            ++failures;
        }
    }
    if (both) {
        discretization_params dp;
        dp.distance_step_count = 20.f;
        dp.angle_step = angle_step;
        mcloud->set_resolution(res);
        model<P> gm(mcloud, dp);
        sample_parameters sp;
        gm.init(sp); /* default = the reference's subset filter, curvature on the GPU */
        if (gm.point_count() != subset.size()) {
            printf("FAIL model<P>::init subset (%u) differs from the oracle's masks (%zu)\n", gm.point_count(), subset.size());
            ++failures;
        }
        tm_scene* gs = nullptr;
        detail::check(tm_scene_create(gm.context(), scloud->points[0].raw(), (uint32_t)scloud->size(), tmask.data(), &gs),
                      "tm_scene_create");
        detail::hip_backend<P> hb;
        hb.s = gs;
        hb.m = gm.handle();
        hb.scloud = scloud.get();
        hb.mcloud = mcloud.get();
        hb.n_t = gm.point_count();
        hb.diam = gm.diameter();
        hb.p = p;
        auto rg = search::find_all(hb, tmask, p);
        dump("hip backend", rg);
        bool same = rg.size() == ro.size();
        for (size_t k = 0; same && k < rg.size(); ++k) {
            same = same && memcmp(rg[k].transform.m, ro[k].transform.m, 64) == 0;
            same = same && rg[k].scene_corrs == ro[k].scene_corrs && rg[k].model_corrs == ro[k].model_corrs;
            same = same && std::fabs(rg[k].signed_score - ro[k].signed_score) <= 1e-9 * std::fabs(ro[k].signed_score);
        }
        printf("%s hip backend == oracle backend (candidates, correspondences, transforms)\n", same ? "ok  " : "FAIL");
        if (!same) ++failures;
        tm_scene_destroy(gs);
    }
    tmo_scene_destroy(ob.s);
    tmo_model_destroy(ob.m);
    printf("%s (%d failures)\n", failures ? "FAILED" : "PASSED", failures);
    return failures ? 1 : 0;
}
