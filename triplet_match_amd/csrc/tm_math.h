/*
 * tm_math.h — scalar building blocks of the triplet-match hot path, usable from HIP device code and
 * from the host side of the C ABI (the reference's free functions feature/discretize/... are host
 * API, include/feature:18-36, include/discretize:14-17).
 *
 * Arithmetic contract: fp32, no FMA contraction (the library is built with -ffp-contract=off and
 * IEEE-correct division / sqrt), fixed association — see DESIGN.md "Numerics". atan2 is a
 * self-contained Cephes-style atanf so that device and host agree bit for bit.
 */
#ifndef TM_MATH_H_
#define TM_MATH_H_

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define TM_HD __host__ __device__ __forceinline__
#else
#define TM_HD inline
#endif

namespace tmx {

struct f3 {
    float x, y, z;
};

TM_HD f3 make3(float x, float y, float z) {
    f3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
TM_HD f3 operator-(f3 a, f3 b) { return make3(a.x - b.x, a.y - b.y, a.z - b.z); }
TM_HD f3 operator+(f3 a, f3 b) { return make3(a.x + b.x, a.y + b.y, a.z + b.z); }
TM_HD f3 scale(f3 a, float s) { return make3(a.x * s, a.y * s, a.z * s); }
TM_HD f3 over(f3 a, float s) { return make3(a.x / s, a.y / s, a.z / s); }
TM_HD float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* Skilling's axes -> transposed Hilbert index (21 bits per axis), in place: after the call the Hilbert key
 * is the bit-interleave of X[0] (most significant), X[1], X[2] */
TM_HD void hilbert_transpose21(unsigned int X[3]) {
    const unsigned int M = 1u << 20;
    for (unsigned int Q = M; Q > 1; Q >>= 1) {
        const unsigned int P = Q - 1;
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const unsigned int t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned int t = 0;
    for (unsigned int Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t;
    X[1] ^= t;
    X[2] ^= t;
}

/* Voxel-table addressing. The table is stored in 4x4x4-voxel bricks (64 consecutive entries); the bricks
 * themselves are grouped 4x4x4 into super-bricks (16^3 voxels, 64 consecutive bricks = 16 KB), which are laid
 * out x-fastest. `bx`, `by` = super-brick counts along x and y. A wave's 8x8-point surface patch (16x16
 * voxels) then touches one to four contiguous 16-KB blocks instead of bricks that are a whole grid row
 * or slab apart. */
TM_HD uint32_t brick_of(uint32_t ix, uint32_t iy, uint32_t iz, uint32_t bx, uint32_t by) {
    return ((((iz >> 4) * by + (iy >> 4)) * bx + (ix >> 4)) << 6) |
           ((((iz >> 2) & 3u) << 4) | (((iy >> 2) & 3u) << 2) | ((ix >> 2) & 3u));
}
/* inverse: brick number -> brick coordinates (units of 4 voxels) */
TM_HD void brick_coords(uint32_t brick, uint32_t bx, uint32_t by, uint32_t& bi, uint32_t& bj, uint32_t& bk) {
    const uint32_t sb = brick >> 6, loc = brick & 63u;
    const uint32_t sk = sb / (bx * by), sr = sb - sk * (bx * by);
    const uint32_t sj = sr / bx, si = sr - sj * bx;
    bi = (si << 2) | (loc & 3u);
    bj = (sj << 2) | ((loc >> 2) & 3u);
    bk = (sk << 2) | (loc >> 4);
}


TM_HD float len2(f3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
TM_HD float len(f3 a) { return sqrtf(len2(a)); }
TM_HD f3 cross3(f3 a, f3 b) {
    return make3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* Eigen normalize(): divide by sqrt(squaredNorm) when it is > 0 */
TM_HD f3 unit(f3 a) {
    float q = len2(a);
    return q > 0.f ? over(a, sqrtf(q)) : a;
}

/* rigid pose, column-major 4x4 with last row (0,0,0,1): only the 12 live entries are kept */
struct pose12 {
    float r00, r10, r20; /* column 0 */
    float r01, r11, r21; /* column 1 */
    float r02, r12, r22; /* column 2 */
    float tx, ty, tz;    /* column 3 */
};
TM_HD pose12 load_pose(const float* m) {
    pose12 p;
    p.r00 = m[0];
    p.r10 = m[1];
    p.r20 = m[2];
    p.r01 = m[4];
    p.r11 = m[5];
    p.r21 = m[6];
    p.r02 = m[8];
    p.r12 = m[9];
    p.r22 = m[10];
    p.tx = m[12];
    p.ty = m[13];
    p.tz = m[14];
    return p;
}
/* T * (x,y,z,1): ((m0*x + m4*y) + m8*z) + m12*1 */
TM_HD f3 xform_point(const pose12& p, f3 v) {
    return make3(((p.r00 * v.x + p.r01 * v.y) + p.r02 * v.z) + p.tx,
                 ((p.r10 * v.x + p.r11 * v.y) + p.r12 * v.z) + p.ty,
                 ((p.r20 * v.x + p.r21 * v.y) + p.r22 * v.z) + p.tz);
}
TM_HD f3 xform_dir(const pose12& p, f3 v) {
    return make3((p.r00 * v.x + p.r01 * v.y) + p.r02 * v.z, (p.r10 * v.x + p.r11 * v.y) + p.r12 * v.z,
                 (p.r20 * v.x + p.r21 * v.y) + p.r22 * v.z);
}

/* ---- MurmurHash3_x86_32 over u32 words (reference include/impl/discretize.hpp:5-45) ---- */
TM_HD uint32_t rotl(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
TM_HD uint32_t murmur_mix(uint32_t h, uint32_t k) {
    k *= 0xcc9e2d51u;
    k = rotl(k, 15);
    k *= 0x1b873593u;
    h ^= k;
    h = rotl(h, 13);
    return h * 5u + 0xe6546b64u;
}
TM_HD uint32_t murmur_fin(uint32_t h, uint32_t nbytes) {
    h ^= nbytes;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
static constexpr uint32_t kMurmurSeed = 42u; /* include/impl/discretize.hpp:10 */
TM_HD uint32_t murmur_key4(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
    uint32_t h = kMurmurSeed;
    h = murmur_mix(h, k0);
    h = murmur_mix(h, k1);
    h = murmur_mix(h, k2);
    h = murmur_mix(h, k3);
    return murmur_fin(h, 16u);
}
TM_HD uint32_t murmur_words(const uint32_t* w, int n, uint32_t seed) {
    uint32_t h = seed;
    for (int i = 0; i < n; ++i) h = murmur_mix(h, w[i]);
    return murmur_fin(h, 4u * (uint32_t)n);
}

/* ---- quantisers (reference src/discretize.cpp:20-30) ---- */
TM_HD uint32_t quantize_range(float v, float lo, float span, uint32_t steps) {
    float t = (v - lo) / span;
    if (t < 0.f) return 0u;
    if (t >= 1.f) return steps - 1u;
    float s = t * (float)steps;
    return (s == s) ? (uint32_t)s : 0u; /* NaN (span == 0) -> 0 */
}
TM_HD uint32_t quantize_step(float v, float step) {
    float q = v / step;
    return (q >= 0.f) ? (uint32_t)q : 0u; /* NaN and negatives -> 0 */
}

/* ---- atan2 ---- */
TM_HD float atan_reduced(float a) {
    float s = 1.f, x = a;
    if (a < 0.f) {
        s = -1.f;
        x = -a;
    }
    float base;
    if (x > 2.414213562373095f) {
        base = 1.57079632679489661923f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        base = 0.78539816339744830962f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        base = 0.f;
    }
    float z = x * x;
    float poly =
        (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) *
            z * x +
        x;
    return s * (base + poly);
}
TM_HD float atan2_def(float y, float x) {
    if (x == 0.f) {
        if (y > 0.f) return 1.57079632679489661923f;
        if (y < 0.f) return -1.57079632679489661923f;
        return 0.f;
    }
    float w = 0.f;
    if (x < 0.f) w = (y < 0.f) ? -3.14159265358979323846f : 3.14159265358979323846f;
    return w + atan_reduced(y / x);
}

/* ---- pair descriptor (reference include/impl/feature.hpp:5-8,16-34) ---- */
struct feat4 {
    float f0, f1, f2, f3;
};
TM_HD float pair_angle(f3 d, f3 t) { return atan2_def(len(cross3(d, t)), fabsf(dot3(d, t))); }
TM_HD feat4 pair_feature(f3 p0, f3 t0, f3 p1, f3 t1) {
    f3 d = p1 - p0;
    feat4 f;
    f.f0 = len(d);
    f.f1 = pair_angle(d, t0);
    f.f2 = pair_angle(d, t1);
    f.f3 = f.f0;
    return f;
}
/* reference include/impl/feature.hpp:50-88: only dim 0 and the two angles are tested */
TM_HD bool feature_valid(const feat4& f, float lo0, float hi0) {
    if (f.f0 < lo0 || f.f0 > hi0) return false;
    const float pi = 3.14159265358979323846f;
    return (f.f1 >= 0.f && f.f1 <= pi) && (f.f2 >= 0.f && f.f2 <= pi);
}
struct key4 {
    uint32_t k0, k1, k2, k3;
};
/* reference include/impl/feature.hpp:38-46: dim 3 uses the dim-0 bounds, so k3 == k0 */
TM_HD key4 feature_key(const feat4& f, float lo0, float hi0, uint32_t steps, float angle_step) {
    float span = hi0 - lo0;
    key4 k;
    k.k0 = quantize_range(f.f0, lo0, span, steps);
    k.k1 = quantize_step(f.f1, angle_step);
    k.k2 = quantize_step(f.f2, angle_step);
    k.k3 = quantize_range(f.f3, lo0, span, steps);
    return k;
}

/* ---- base transform (reference include/impl/scene.hpp:538-567): maps scene -> model ---- */
TM_HD void base_transform(f3 src_i, f3 src_j, f3 src_t, f3 tgt_i, f3 tgt_j, f3 tgt_t, float* T) {
    f3 ua = unit(src_j - src_i);
    f3 ub = unit(tgt_j - tgt_i);
    f3 va = src_t - scale(ua, dot3(src_t, ua));
    f3 vb = tgt_t - scale(ub, dot3(tgt_t, ub));
    va = unit(va);
    vb = unit(vb);
    f3 wa = unit(cross3(ua, va));
    f3 wb = unit(cross3(ub, vb));
    /* A = [ua va wa] (columns); inverse = adjugate * (1/det), Eigen's 3x3 path */
    float a00 = ua.x, a01 = va.x, a02 = wa.x;
    float a10 = ua.y, a11 = va.y, a12 = wa.y;
    float a20 = ua.z, a21 = va.z, a22 = wa.z;
    float c00 = a11 * a22 - a12 * a21; /* cofactor(0,0) */
    float c10 = a21 * a02 - a22 * a01; /* cofactor(1,0) */
    float c20 = a01 * a12 - a02 * a11; /* cofactor(2,0) */
    float c01 = a12 * a20 - a10 * a22; /* cofactor(0,1) */
    float c11 = a22 * a00 - a20 * a02;
    float c21 = a02 * a10 - a00 * a12;
    float c02 = a10 * a21 - a11 * a20;
    float c12 = a20 * a01 - a21 * a00;
    float c22 = a00 * a11 - a01 * a10;
    float det = (c00 * a00 + c10 * a10) + c20 * a20;
    float id = 1.0f / det;
    /* inv(r,c) = cofactor(c,r) * id */
    float i00 = c00 * id, i01 = c10 * id, i02 = c20 * id;
    float i10 = c01 * id, i11 = c11 * id, i12 = c21 * id;
    float i20 = c02 * id, i21 = c12 * id, i22 = c22 * id;
    float b00 = ub.x, b01 = vb.x, b02 = wb.x;
    float b10 = ub.y, b11 = vb.y, b12 = wb.y;
    float b20 = ub.z, b21 = vb.z, b22 = wb.z;
    float r00 = (b00 * i00 + b01 * i10) + b02 * i20;
    float r01 = (b00 * i01 + b01 * i11) + b02 * i21;
    float r02 = (b00 * i02 + b01 * i12) + b02 * i22;
    float r10 = (b10 * i00 + b11 * i10) + b12 * i20;
    float r11 = (b10 * i01 + b11 * i11) + b12 * i21;
    float r12 = (b10 * i02 + b11 * i12) + b12 * i22;
    float r20 = (b20 * i00 + b21 * i10) + b22 * i20;
    float r21 = (b20 * i01 + b21 * i11) + b22 * i21;
    float r22 = (b20 * i02 + b21 * i12) + b22 * i22;
    float ox = (r00 * src_i.x + r01 * src_i.y) + r02 * src_i.z;
    float oy = (r10 * src_i.x + r11 * src_i.y) + r12 * src_i.z;
    float oz = (r20 * src_i.x + r21 * src_i.y) + r22 * src_i.z;
    T[0] = r00;
    T[1] = r10;
    T[2] = r20;
    T[3] = 0.f;
    T[4] = r01;
    T[5] = r11;
    T[6] = r21;
    T[7] = 0.f;
    T[8] = r02;
    T[9] = r12;
    T[10] = r22;
    T[11] = 0.f;
    T[12] = tgt_i.x - ox;
    T[13] = tgt_i.y - oy;
    T[14] = tgt_i.z - oz;
    T[15] = 1.f;
}

/* ---- early-drop bound, positive form of reference include/impl/scene.hpp:493-498 ---- */
TM_HD uint32_t early_drop_upper(uint32_t tried, uint32_t subset, uint32_t corrs) {
    double n = 2.0 + (double)tried, x = 2.0 + (double)subset, c = 1.0 + (double)corrs;
    double root = sqrt((x * c * (x - n) * (n - c)) / (n + 1.0));
    uint32_t t = (uint32_t)((x * c + root) / n);
    return t - 1u;
}

/* ---- symmetric 3x3 eigen-decomposition: cyclic Jacobi, fixed 10 sweeps, f32 (the library's definition of
 * the reference's pcl::eigen33 call, include/impl/pointcloud.hpp:38-40). a = (a00,a01,a02,a11,a12,a22).
 * Eigenvalues ascending in ev[3]; vmax = unit eigenvector of the largest. ---- */
TM_HD void jacobi_eigen3(const float a_in[6], float ev[3], float vmax[3]) {
    float a[3][3] = {{a_in[0], a_in[1], a_in[2]}, {a_in[1], a_in[3], a_in[4]}, {a_in[2], a_in[4], a_in[5]}};
    float v[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
    for (int sweep = 0; sweep < 10; ++sweep) {
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2, r = 3 - p - q;
            const float apq = a[p][q];
            if (apq == 0.f) continue;
            const float theta = (a[q][q] - a[p][p]) / (2.f * apq);
            const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
            const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
            a[p][p] = a[p][p] - t * apq;
            a[q][q] = a[q][q] + t * apq;
            a[p][q] = 0.f;
            a[q][p] = 0.f;
            const float arp = a[r][p], arq = a[r][q];
            a[r][p] = c * arp - s * arq;
            a[p][r] = a[r][p];
            a[r][q] = s * arp + c * arq;
            a[q][r] = a[r][q];
            for (int k = 0; k < 3; ++k) {
                const float vkp = v[k][p], vkq = v[k][q];
                v[k][p] = c * vkp - s * vkq;
                v[k][q] = s * vkp + c * vkq;
            }
        }
    }
    int i0 = 0, i1 = 1, i2 = 2; /* sort indices by eigenvalue, ascending */
    if (a[i0][i0] > a[i1][i1]) { int t = i0; i0 = i1; i1 = t; }
    if (a[i1][i1] > a[i2][i2]) { int t = i1; i1 = i2; i2 = t; }
    if (a[i0][i0] > a[i1][i1]) { int t = i0; i0 = i1; i1 = t; }
    ev[0] = a[i0][i0];
    ev[1] = a[i1][i1];
    ev[2] = a[i2][i2];
    vmax[0] = v[0][i2];
    vmax[1] = v[1][i2];
    vmax[2] = v[2][i2];
}

/* voxel grid affine: v = s*p + t per axis; a hit is trunc(v) in [0, ext) which, for the float v,
 * is v > -1 && v < ext (reference include/impl/model.hpp:180-192; NaN fails both) */
struct grid_t {
    float sx, sy, sz;
    float tx, ty, tz;
    float fex, fey, fez; /* extents as float */
    int ex, ey, ez;
};

}  // namespace tmx

#endif /* TM_MATH_H_ */
