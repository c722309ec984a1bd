// cylinder_traits::init_from_model (the restated MSAC cylinder fit, include/triplet_match/traits.hpp; reference
// include/impl/cylinder_traits.hpp:5-50): reads a cloud (n, then n x 12 floats) and a resolution, writes the fitted
// state (radius, threshold, origin[3], axis[3]) for the Python side to judge with its own arithmetic. Host only.
#include <cstdio>
#include <fstream>
#include <vector>

#include "triplet_match/cylinder_traits.hpp"

using namespace triplet_match;
typedef PointSurfel P;

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    uint32_t n = 0, n_subset = 0;
    float res = 0.f;
    in.read((char*)&n, 4);
    auto cloud = pointcloud<P>::empty();
    cloud->points.resize(n);
    in.read((char*)cloud->points.data(), (std::streamsize)n * 48);
    in.read((char*)&res, 4);
    in.read((char*)&n_subset, 4);
    std::vector<uint32_t> sub(n_subset);
    if (n_subset) in.read((char*)sub.data(), (std::streamsize)n_subset * 4);
    if (!in.good()) return 3;
    cloud->set_resolution(res);
    subset_t subset(sub.begin(), sub.end());
    const uint32_t seed = (uint32_t)atoi(argv[3]);
    cylinder_traits<P>::handle_t h;
    try {
        h = cylinder_traits<P>::init_from_model(cloud, subset, seed);
    } catch (const std::exception& e) {
        std::printf("threw: %s\n", e.what());
        return 4;
    }
    const float out[8] = {h->radius, h->threshold, h->origin[0], h->origin[1], h->origin[2], h->g2l(2, 0), h->g2l(2, 1), h->g2l(2, 2)};
    std::ofstream o(argv[2], std::ios::binary);
    o.write((const char*)out, sizeof out);
    /* the frame is usable: a surface point projects to w ~ 0, unproject(project(x)) == x */
    int bad = 0;
    for (uint32_t i = 0; i < n; i += 97) {
        const P& p = cloud->points[i];
        auto uvw = cylinder_traits<P>::project(h, vec3f_t(p.x, p.y, p.z));
        if (!uvw) continue;
        vec3f_t back = cylinder_traits<P>::unproject(h, *uvw);
        if ((back - vec3f_t(p.x, p.y, p.z)).norm() > 1e-4f * (1.f + vec3f_t(p.x, p.y, p.z).norm())) ++bad;
    }
    std::printf("radius %.5f threshold %.5f round-trip failures %d\n", h->radius, h->threshold, bad);
    return bad ? 5 : 0;
}
