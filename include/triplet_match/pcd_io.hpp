/*
 * triplet_match/pcd_io.hpp — PCD (Point Cloud Data v0.7) reader/writer for the 48-byte surfel record.
 *
 * Replaces what pointcloud<Point>::from_pcd gets from pcl::io::loadPCDFile (reference
 * include/impl/pointcloud.hpp:58-64; PCL itself is not part of the reference tree). Written from the
 * published file-format description: a text header (VERSION, FIELDS, SIZE, TYPE, COUNT, WIDTH, HEIGHT,
 * VIEWPOINT, POINTS, DATA) followed by
 *   DATA ascii              one point per line, fields separated by blanks;
 *   DATA binary             POINTS packed records, fields in header order;
 *   DATA binary_compressed  u32 compressed size, u32 uncompressed size, LZF stream of the records
 *                           re-ordered field by field (all x, then all y, ...).
 * Fields are matched by name onto the surfel: x y z | normal_x normal_y normal_z | rgba (or rgb) radius
 * confidence curvature — the reference keeps the crease tangent in the last three (include/common:62-70).
 * Unknown fields (including "_" padding) are skipped; missing ones keep the surfel defaults
 * (w = 1, everything else 0). Host-side only; nothing here touches the GPU.
 */
#ifndef TRIPLET_MATCH_PCD_IO_HPP_
#define TRIPLET_MATCH_PCD_IO_HPP_

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "common.hpp"

namespace triplet_match {
namespace pcd {

enum class data_mode { ascii, binary, binary_compressed };

namespace detail {

struct field_t {
    std::string name;
    uint32_t size = 4;
    char type = 'F';
    uint32_t count = 1;
    uint32_t offset = 0; /* byte offset inside one packed record */
    int slot = -1;       /* float slot 0..11 of the surfel, -1 = skipped */
};

inline int surfel_slot(const std::string& name) {
    static const struct { const char* n; int s; } map[] = {
        {"x", 0}, {"y", 1}, {"z", 2}, {"normal_x", 4}, {"normal_y", 5}, {"normal_z", 6},
        {"rgba", 8}, {"rgb", 8}, {"radius", 9}, {"confidence", 10}, {"curvature", 11}};
    for (const auto& m : map)
        if (name == m.n) return m.s;
    return -1;
}

/* one binary value -> the 4 bytes stored in the surfel slot (float, or the raw colour word) */
inline uint32_t convert_binary(const unsigned char* p, const field_t& f) {
    const bool colour = f.slot == 8;
    auto from_float = [&](float v) {
        uint32_t b;
        std::memcpy(&b, &v, 4);
        return b;
    };
    if (f.type == 'F') {
        if (f.size == 4) {
            uint32_t b;
            std::memcpy(&b, p, 4);
            return b;
        }
        if (f.size == 8) {
            double d;
            std::memcpy(&d, p, 8);
            return from_float((float)d);
        }
        throw std::runtime_error("pcd: unsupported float size");
    }
    uint64_t u = 0;
    int64_t s = 0;
    switch (f.size) {
        case 1: u = *p; s = (int8_t)*p; break;
        case 2: { uint16_t t; std::memcpy(&t, p, 2); u = t; s = (int16_t)t; } break;
        case 4: { uint32_t t; std::memcpy(&t, p, 4); u = t; s = (int32_t)t; } break;
        case 8: { uint64_t t; std::memcpy(&t, p, 8); u = t; s = (int64_t)t; } break;
        default: throw std::runtime_error("pcd: unsupported integer size");
    }
    if (colour) return (uint32_t)u;
    return from_float(f.type == 'I' ? (float)s : (float)u);
}

inline uint32_t convert_ascii(const std::string& tok, const field_t& f) {
    const bool colour = f.slot == 8;
    if (colour && f.type != 'F') return (uint32_t)std::strtoull(tok.c_str(), nullptr, 10);
    float v;
    if (f.type == 'F') {
        v = std::strtof(tok.c_str(), nullptr); /* accepts nan / inf as PCL's reader does */
    } else if (f.type == 'I') {
        v = (float)std::strtoll(tok.c_str(), nullptr, 10);
    } else {
        v = (float)std::strtoull(tok.c_str(), nullptr, 10);
    }
    uint32_t b;
    std::memcpy(&b, &v, 4);
    return b;
}

/* LZF decompression (Lehmann's format as used by PCD): control byte c; c < 32 -> c+1 literal bytes;
 * otherwise a back reference of length (c>>5)+2 (a length field of 7 is extended by the next byte) at
 * distance ((c&31)<<8 | next byte) + 1. Returns false on a malformed stream. */
inline bool lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        uint32_t c = in[ip++];
        if (c < 32) {
            size_t run = c + 1;
            if (ip + run > in_len || op + run > out_len) return false;
            std::memcpy(out + op, in + ip, run);
            ip += run;
            op += run;
        } else {
            size_t len = c >> 5;
            if (len == 7) {
                if (ip >= in_len) return false;
                len += in[ip++];
            }
            if (ip >= in_len) return false;
            size_t dist = ((size_t)(c & 31) << 8 | in[ip++]) + 1;
            len += 2;
            if (dist > op || op + len > out_len) return false;
            for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - dist]; /* may overlap */
        }
    }
    return op == out_len;
}

/* LZF compression: greedy, 3-byte hash, matches up to 264 bytes at distances up to 8192. */
inline std::vector<unsigned char> lzf_compress(const unsigned char* in, size_t n) {
    std::vector<unsigned char> out;
    out.reserve(n + n / 32 + 8);
    std::vector<int64_t> table(1 << 16, -1);
    size_t lit_start = 0, ip = 0;
    auto flush_literals = [&](size_t end) {
        while (lit_start < end) {
            size_t run = std::min<size_t>(32, end - lit_start);
            out.push_back((unsigned char)(run - 1));
            out.insert(out.end(), in + lit_start, in + lit_start + run);
            lit_start += run;
        }
    };
    while (ip + 2 < n) {
        uint32_t h = ((uint32_t)in[ip] << 16 | (uint32_t)in[ip + 1] << 8 | in[ip + 2]) * 2654435761u >> 16;
        int64_t ref = table[h];
        table[h] = (int64_t)ip;
        if (ref >= 0 && ip - (size_t)ref <= 8192 && in[ref] == in[ip] && in[ref + 1] == in[ip + 1] &&
            in[ref + 2] == in[ip + 2]) {
            size_t len = 3;
            while (len < 264 && ip + len < n && in[ref + len] == in[ip + len]) ++len;
            flush_literals(ip);
            size_t dist = ip - (size_t)ref - 1, l = len - 2;
            if (l < 7) {
                out.push_back((unsigned char)((l << 5) | (dist >> 8)));
            } else {
                out.push_back((unsigned char)((7u << 5) | (dist >> 8)));
                out.push_back((unsigned char)(l - 7));
            }
            out.push_back((unsigned char)(dist & 255));
            ip += len;
            lit_start = ip;
        } else {
            ++ip;
        }
    }
    flush_literals(n);
    return out;
}

}  // namespace detail

/* Reads a PCD file into 48-byte surfels. Throws std::runtime_error on I/O or format errors. */
template <typename Point>
inline void read(const std::string& filename, std::vector<Point>& points) {
    static_assert(sizeof(Point) == 48, "Point must be the 48-byte surfel layout");
    std::ifstream in(filename, std::ios::binary);
    if (!in) throw std::runtime_error("pcd: could not open " + filename);
    std::vector<detail::field_t> fields;
    uint64_t width = 0, height = 1, n_points = 0;
    bool have_points = false, have_data = false;
    data_mode mode = data_mode::ascii;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string key;
        if (!(ls >> key) || key[0] == '#') continue;
        auto fill = [&](auto setter) {
            std::string tok;
            size_t i = 0;
            while (ls >> tok) {
                if (i >= fields.size()) throw std::runtime_error("pcd: header line longer than FIELDS");
                setter(fields[i++], tok);
            }
        };
        if (key == "VERSION" || key == "VIEWPOINT") {
            continue;
        } else if (key == "FIELDS" || key == "COLUMNS") {
            std::string tok;
            while (ls >> tok) {
                detail::field_t f;
                f.name = tok;
                fields.push_back(f);
            }
        } else if (key == "SIZE") {
            fill([](detail::field_t& f, const std::string& t) { f.size = (uint32_t)std::stoul(t); });
        } else if (key == "TYPE") {
            fill([](detail::field_t& f, const std::string& t) { f.type = t[0]; });
        } else if (key == "COUNT") {
            fill([](detail::field_t& f, const std::string& t) { f.count = (uint32_t)std::stoul(t); });
        } else if (key == "WIDTH") {
            ls >> width;
        } else if (key == "HEIGHT") {
            ls >> height;
        } else if (key == "POINTS") {
            ls >> n_points;
            have_points = true;
        } else if (key == "DATA") {
            std::string m;
            ls >> m;
            if (m == "ascii") mode = data_mode::ascii;
            else if (m == "binary") mode = data_mode::binary;
            else if (m == "binary_compressed") mode = data_mode::binary_compressed;
            else throw std::runtime_error("pcd: unknown DATA mode '" + m + "'");
            have_data = true;
            break;
        } else {
            throw std::runtime_error("pcd: unknown header entry '" + key + "'");
        }
    }
    if (!have_data || fields.empty()) throw std::runtime_error("pcd: incomplete header in " + filename);
    if (!have_points) n_points = width * height;
    uint32_t record = 0;
    for (auto& f : fields) {
        if (f.type != 'F' && f.type != 'I' && f.type != 'U') throw std::runtime_error("pcd: bad TYPE");
        if (f.size == 0 || f.size > 8) throw std::runtime_error("pcd: bad SIZE");
        f.offset = record;
        record += f.size * f.count;
        f.slot = f.count >= 1 ? detail::surfel_slot(f.name) : -1;
    }
    points.assign((size_t)n_points, Point());
    if (n_points == 0) return;
    auto slots = [&](size_t i) { return reinterpret_cast<uint32_t*>(&points[i]); };

    if (mode == data_mode::ascii) {
        for (size_t i = 0; i < n_points; ++i) {
            if (!std::getline(in, line)) throw std::runtime_error("pcd: fewer points than POINTS says");
            std::istringstream ls(line);
            std::string tok;
            for (const auto& f : fields)
                for (uint32_t c = 0; c < f.count; ++c) {
                    if (!(ls >> tok)) throw std::runtime_error("pcd: short ascii record");
                    if (c == 0 && f.slot >= 0) slots(i)[f.slot] = detail::convert_ascii(tok, f);
                }
        }
        return;
    }
    std::vector<unsigned char> raw((size_t)n_points * record);
    if (mode == data_mode::binary) {
        in.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)raw.size());
        if ((size_t)in.gcount() != raw.size()) throw std::runtime_error("pcd: truncated binary data");
        for (size_t i = 0; i < n_points; ++i)
            for (const auto& f : fields)
                if (f.slot >= 0) slots(i)[f.slot] = detail::convert_binary(&raw[i * record + f.offset], f);
        return;
    }
    uint32_t sizes[2];
    in.read(reinterpret_cast<char*>(sizes), 8);
    if (in.gcount() != 8) throw std::runtime_error("pcd: truncated compressed header");
    if (sizes[1] != raw.size()) throw std::runtime_error("pcd: uncompressed size does not match the header");
    std::vector<unsigned char> comp(sizes[0]);
    in.read(reinterpret_cast<char*>(comp.data()), (std::streamsize)comp.size());
    if ((size_t)in.gcount() != comp.size()) throw std::runtime_error("pcd: truncated compressed data");
    if (!detail::lzf_decompress(comp.data(), comp.size(), raw.data(), raw.size()))
        throw std::runtime_error("pcd: corrupt LZF stream");
    size_t base = 0; /* field-major: n_points values of field 0, then field 1, ... */
    for (const auto& f : fields) {
        const size_t stride = (size_t)f.size * f.count;
        if (f.slot >= 0)
            for (size_t i = 0; i < n_points; ++i) slots(i)[f.slot] = detail::convert_binary(&raw[base + i * stride], f);
        base += stride * n_points;
    }
}

/* Writes the ten surfel fields PCL declares for PointSurfel (40 bytes per point in the binary modes). */
template <typename Point>
inline void write(const std::string& filename, const std::vector<Point>& points,
                  data_mode mode = data_mode::binary) {
    static_assert(sizeof(Point) == 48, "Point must be the 48-byte surfel layout");
    static const int slot_of[10] = {0, 1, 2, 4, 5, 6, 8, 9, 10, 11};
    std::ofstream out(filename, std::ios::binary);
    if (!out) throw std::runtime_error("pcd: could not create " + filename);
    const size_t n = points.size();
    out << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
        << "FIELDS x y z normal_x normal_y normal_z rgba radius confidence curvature\n"
        << "SIZE 4 4 4 4 4 4 4 4 4 4\nTYPE F F F F F F U F F F\nCOUNT 1 1 1 1 1 1 1 1 1 1\n"
        << "WIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA "
        << (mode == data_mode::ascii ? "ascii" : mode == data_mode::binary ? "binary" : "binary_compressed")
        << "\n";
    auto word = [&](size_t i, int f) { return reinterpret_cast<const uint32_t*>(&points[i])[slot_of[f]]; };
    if (mode == data_mode::ascii) {
        char buf[64];
        for (size_t i = 0; i < n; ++i) {
            for (int f = 0; f < 10; ++f) {
                uint32_t b = word(i, f);
                if (f == 6) {
                    std::snprintf(buf, sizeof buf, "%u", b);
                } else {
                    float v;
                    std::memcpy(&v, &b, 4);
                    std::snprintf(buf, sizeof buf, "%.9g", v); /* 9 digits round-trip an fp32 */
                }
                out << buf << (f == 9 ? '\n' : ' ');
            }
        }
    } else if (mode == data_mode::binary) {
        std::vector<uint32_t> rec(10 * n);
        for (size_t i = 0; i < n; ++i)
            for (int f = 0; f < 10; ++f) rec[10 * i + f] = word(i, f);
        out.write(reinterpret_cast<const char*>(rec.data()), (std::streamsize)(rec.size() * 4));
    } else {
        std::vector<uint32_t> soa(10 * n);
        for (int f = 0; f < 10; ++f)
            for (size_t i = 0; i < n; ++i) soa[(size_t)f * n + i] = word(i, f);
        auto comp = detail::lzf_compress(reinterpret_cast<const unsigned char*>(soa.data()), soa.size() * 4);
        uint32_t sizes[2] = {(uint32_t)comp.size(), (uint32_t)(soa.size() * 4)};
        out.write(reinterpret_cast<const char*>(sizes), 8);
        out.write(reinterpret_cast<const char*>(comp.data()), (std::streamsize)comp.size());
    }
    if (!out) throw std::runtime_error("pcd: write failed for " + filename);
}

}  // namespace pcd
}  // namespace triplet_match

#endif
