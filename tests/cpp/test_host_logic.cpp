/*
 * test_host_logic.cpp — the host-side pieces of the search that the reference takes from third parties or
 * writes inline, checked against INDEPENDENT answers (no oracle, no GPU):
 *   rigid_fit <in> <out>  search::detail::rigid_fit (Horn) on correspondence sets read from <in>; the
 *                         poses go to <out> and are compared by pytest with numpy SVD-Umeyama vectors
 *                         (tests/golden/icp_sets.npz). Reference: Eigen::umeyama(scn, mdl, false),
 *                         include/impl/scene.hpp:385-393.
 *   arbitrate             search::arbitrate on hand-built overlapping candidates whose accepted set was
 *                         worked out by hand. Reference: include/impl/scene.hpp:193-237.
 * search.hpp is header-only host code; nothing here touches libtm_hip.so.
 */
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>

#include "triplet_match/search.hpp"

using namespace triplet_match;

static int run_rigid_fit(const char* in_path, const char* out_path) {
    std::ifstream in(in_path, std::ios::binary);
    std::ofstream out(out_path, std::ios::binary);
    uint32_t n_sets = 0;
    in.read((char*)&n_sets, 4);
    for (uint32_t s = 0; s < n_sets; ++s) {
        uint32_t n = 0;
        in.read((char*)&n, 4);
        std::vector<float> sc(3 * (size_t)n), md(3 * (size_t)n);
        in.read((char*)sc.data(), (std::streamsize)sc.size() * 4);
        in.read((char*)md.data(), (std::streamsize)md.size() * 4);
        if (!in.good()) return 3;
        search::detail::icp_sums_t acc;
        for (uint32_t k = 0; k < n; ++k) search::detail::icp_sums_add(acc, &sc[3 * (size_t)k], &md[3 * (size_t)k]);
        search::pose_t T = search::detail::rigid_fit(acc);
        out.write((const char*)T.m, 64);
    }
    return out.good() ? 0 : 4;
}

static int failures = 0;
#define CHECK(c, what)                               \
    do {                                             \
        bool _ok = (c);                              \
        printf("%s %s\n", _ok ? "ok  " : "FAIL", what); \
        if (!_ok) ++failures;                        \
    } while (0)

static search::candidate_t cand(int id, std::initializer_list<uint32_t> pts) {
    search::candidate_t c;
    for (int k = 0; k < 16; ++k) c.transform.m[k] = (float)id;
    c.signed_score = 0.25 * id;
    int k = 0;
    for (uint32_t p : pts) {
        c.scene_corrs.push_back(p);
        c.model_corrs.push_back(100u * (uint32_t)id + (uint32_t)k++);
    }
    return c;
}

static int run_arbitrate() {
    /* 20 scene points, even indices are tangent points; min_points = 2 tangent inliers, min_points_all = 4.
     *   A = {0..5}          B = {4..10}        C = {0,1,11,12,13}   D = {11,13,15,17}   E = {11,12,14,16}
     * order by size (stable): B(7) A(6) C(5) D(4) E(4)
     *   B: all 7 free, tangent 4,6,8,10 = 4 >= 2, all 7 >= 4                       -> accepted, holds 4..10
     *   A: 4,5 taken -> {0,1,2,3}: tangent 0,2 = 2 >= 2, all 4 >= 4                 -> accepted (trimmed)
     *   C: 0,1 taken -> {11,12,13}: all 3 < 4                                      -> rejected, gives 11,12,13 back
     *   D: {11,13,15,17}: tangent 0 < 2                                            -> rejected, gives them back
     *   E: {11,12,14,16} all free again: tangent 12,14,16 = 3, all 4               -> accepted
     * (had C kept 11 and 12, E would be left with 2 points and fail) */
    std::vector<int32_t> tmask(20);
    for (int i = 0; i < 20; ++i) tmask[i] = (i % 2 == 0) ? 1 : 0;
    std::vector<search::candidate_t> cands;
    cands.push_back(cand(1, {0, 1, 2, 3, 4, 5}));
    cands.push_back(cand(2, {4, 5, 6, 7, 8, 9, 10}));
    cands.push_back(cand(3, {0, 1, 11, 12, 13}));
    cands.push_back(cand(4, {11, 13, 15, 17}));
    cands.push_back(cand(5, {11, 12, 14, 16}));
    auto r = search::arbitrate(cands, 20, tmask, 2, 4);
    CHECK(r.size() == 3, "three candidates accepted");
    if (r.size() == 3) {
        CHECK(r[0].transform.m[0] == 2.f && r[0].scene_corrs == std::vector<uint32_t>({4, 5, 6, 7, 8, 9, 10}),
              "largest candidate first, untouched");
        CHECK(r[0].model_corrs == std::vector<uint32_t>({200, 201, 202, 203, 204, 205, 206}), "its model side follows");
        CHECK(r[1].transform.m[0] == 1.f && r[1].scene_corrs == std::vector<uint32_t>({0, 1, 2, 3}),
              "second candidate loses the points the first one holds");
        CHECK(r[1].model_corrs == std::vector<uint32_t>({100, 101, 102, 103}), "model correspondences trimmed alike");
        CHECK(r[2].transform.m[0] == 5.f && r[2].scene_corrs == std::vector<uint32_t>({11, 12, 14, 16}),
              "rejected candidates gave their points back");
        CHECK(r[1].signed_score == 0.25 && r[2].signed_score == 1.25, "signed_score carried through unchanged");
    }
    /* thresholds are inclusive lower bounds (score < min rejects, :227) */
    auto r2 = search::arbitrate({cand(1, {0, 2, 4, 6})}, 20, tmask, 4, 4);
    CHECK(r2.size() == 1, "score == min_points and score_all == min_points_all is accepted");
    auto r3 = search::arbitrate({cand(1, {0, 2, 4, 6})}, 20, tmask, 5, 4);
    CHECK(r3.empty(), "one tangent inlier short is rejected");
    /* equal sizes keep their input order (stable sort; the reference's std::sort leaves it unspecified) */
    auto r4 = search::arbitrate({cand(7, {0, 2, 4, 6}), cand(8, {0, 2, 4, 6})}, 20, tmask, 2, 4);
    CHECK(r4.size() == 1 && r4[0].transform.m[0] == 7.f, "of two identical candidates the first wins, the second is emptied");
    /* more candidates than a byte numbers (the wide owner marks): 300 one-point candidates over 150 points, every
     * point offered twice — the first 150 are accepted, their twins find the point held */
    {
        std::vector<search::candidate_t> many;
        std::vector<int32_t> all_tangent(150, 1);
        for (int k = 0; k < 300; ++k) many.push_back(cand(k, {(uint32_t)(k % 150)}));
        auto rm = search::arbitrate(many, 150, all_tangent, 1, 1);
        bool ok = rm.size() == 150;
        for (size_t k = 0; k < rm.size() && ok; ++k) ok = rm[k].transform.m[0] == (float)k && rm[k].scene_corrs == std::vector<uint32_t>({(uint32_t)k});
        CHECK(ok, "300 candidates: the first of every pair is accepted");
    }
    printf("%s (%d failures)\n", failures ? "FAILED" : "PASSED", failures);
    return failures ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && std::string(argv[1]) == "rigid_fit") return run_rigid_fit(argv[2], argv[3]);
    if (argc >= 2 && std::string(argv[1]) == "arbitrate") return run_arbitrate();
    return 2;
}
