// PCD reader/writer of the drop-in API (include/triplet_match/pcd_io.hpp). Host-only.
//   test_pcd dump <in.pcd> <out.raw>          read, write the 48-byte records
//   test_pcd roundtrip <dir>                  write/read all three DATA modes, compare bit for bit
//   test_pcd from_pcd <in.pcd>                print the size pointcloud::from_pcd returns
#include <cstdio>
#include <cstring>
#include <random>

#include <triplet_match/pointcloud.hpp>

using namespace triplet_match;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::string cmd = argv[1];
    if (cmd == "dump" && argc == 4) {
        std::vector<PointSurfel> pts;
        try {
            pcd::read(argv[2], pts);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 3;
        }
        FILE* f = std::fopen(argv[3], "wb");
        if (!f) return 4;
        std::fwrite(pts.data(), 48, pts.size(), f);
        std::fclose(f);
        return 0;
    }
    if (cmd == "from_pcd") {
        auto c = pointcloud<PointSurfel>::from_pcd(argv[2]);
        std::printf("%zu\n", c->size());
        return 0;
    }
    if (cmd == "roundtrip") {
        std::mt19937 rng(3);
        std::uniform_real_distribution<float> u(-10.f, 10.f);
        std::vector<PointSurfel> pts(5000);
        for (size_t i = 0; i < pts.size(); ++i) {
            auto& p = pts[i];
            p.x = u(rng); p.y = u(rng); p.z = (float)(i % 7);  // repetitive column -> back references
            p.normal_x = 0.f; p.normal_y = 0.f; p.normal_z = 1.f;
            p.rgba = 0xff000000u | (uint32_t)i;
            set_tangent(p, i % 25 == 0 ? vec3f_t(1.f, 0.f, 0.f) : vec3f_t(0.f, 0.f, 0.f));
        }
        pts[17].x = 1e-30f; pts[18].y = -3.4e38f; pts[19].z = 0.1f;
        const pcd::data_mode modes[3] = {pcd::data_mode::ascii, pcd::data_mode::binary,
                                         pcd::data_mode::binary_compressed};
        const char* names[3] = {"ascii", "binary", "binary_compressed"};
        for (int m = 0; m < 3; ++m) {
            std::string path = std::string(argv[2]) + "/rt_" + names[m] + ".pcd";
            pcd::write(path, pts, modes[m]);
            std::vector<PointSurfel> back;
            pcd::read(path, back);
            if (back.size() != pts.size() || std::memcmp(back.data(), pts.data(), 48 * pts.size()) != 0) {
                std::fprintf(stderr, "round trip differs in mode %s\n", names[m]);
                return 1;
            }
        }
        // empty cloud
        std::vector<PointSurfel> none, back;
        pcd::write(std::string(argv[2]) + "/empty.pcd", none, pcd::data_mode::binary_compressed);
        pcd::read(std::string(argv[2]) + "/empty.pcd", back);
        if (!back.empty()) return 1;
        std::printf("ok\n");
        return 0;
    }
    return 2;
}
