/*
 * tm_traits_math.h — projector math of the four traits plugins (identity / plane / plane2 /
 * cylinder), host + device. Replaces the statics of the reference's include/impl/<kind>_traits.hpp.
 * state layout (TM_STATE_FLOATS = 37): [radius, threshold, origin xyz, g2l 16 col-major, l2g 16].
 */
#ifndef TM_TRAITS_MATH_H_
#define TM_TRAITS_MATH_H_

#include "tm_math.h"

namespace tmx {

struct frame_t {
    float radius, threshold;
    f3 origin;
    /* g2l rows (rotation) and translation; l2g = [R^T | origin] */
    f3 r0, r1, r2;
    f3 t;
};

TM_HD frame_t frame_load(const float* st) {
    frame_t f;
    f.radius = st[0];
    f.threshold = st[1];
    f.origin = make3(st[2], st[3], st[4]);
    const float* g = st + 5; /* column-major: g[c*4+r] */
    f.r0 = make3(g[0], g[4], g[8]);
    f.r1 = make3(g[1], g[5], g[9]);
    f.r2 = make3(g[2], g[6], g[10]);
    f.t = make3(g[12], g[13], g[14]);
    return f;
}

TM_HD void frame_store(f3 r0, f3 r1, f3 r2, f3 origin, float radius, float threshold, float* st) {
    st[0] = radius;
    st[1] = threshold;
    st[2] = origin.x;
    st[3] = origin.y;
    st[4] = origin.z;
    float* g = st + 5;
    float* l = st + 21;
    for (int i = 0; i < 16; ++i) g[i] = l[i] = 0.f;
    g[0] = r0.x; g[4] = r0.y; g[8] = r0.z;
    g[1] = r1.x; g[5] = r1.y; g[9] = r1.z;
    g[2] = r2.x; g[6] = r2.y; g[10] = r2.z;
    f3 no = make3(-origin.x, -origin.y, -origin.z);
    g[12] = dot3(r0, no);
    g[13] = dot3(r1, no);
    g[14] = dot3(r2, no);
    g[15] = 1.f;
    /* inverse of an orthonormal frame: transpose | origin */
    l[0] = r0.x; l[1] = r0.y; l[2] = r0.z;
    l[4] = r1.x; l[5] = r1.y; l[6] = r1.z;
    l[8] = r2.x; l[9] = r2.y; l[10] = r2.z;
    l[12] = origin.x;
    l[13] = origin.y;
    l[14] = origin.z;
    l[15] = 1.f;
}

/* g2l * (p,1): ((r.x*p.x + r.y*p.y) + r.z*p.z) + t */
TM_HD f3 frame_point(const frame_t& f, f3 p) {
    return make3(dot3(f.r0, p) + f.t.x, dot3(f.r1, p) + f.t.y, dot3(f.r2, p) + f.t.z);
}
TM_HD f3 frame_dir(const frame_t& f, f3 v) { return make3(dot3(f.r0, v), dot3(f.r1, v), dot3(f.r2, v)); }

/* Eigen unitOrthogonal() for 3-vectors, dummy_precision<float> = 1e-5 */
TM_HD f3 unit_orthogonal(f3 s) {
    bool x_small = fabsf(s.x) <= fabsf(s.z) * 1e-5f;
    bool y_small = fabsf(s.y) <= fabsf(s.z) * 1e-5f;
    if (!x_small || !y_small) {
        float inv = 1.f / sqrtf(s.x * s.x + s.y * s.y);
        return make3(-s.y * inv, s.x * inv, 0.f);
    }
    float inv = 1.f / sqrtf(s.y * s.y + s.z * s.z);
    return make3(0.f, -s.z * inv, s.y * inv);
}

/* reference include/impl/cylinder_traits.hpp:142-145: Jacobian of the unrolling on a direction */
TM_HD f3 cylinder_jacobian(float r, f3 loc, f3 lt) {
    float n = sqrtf(loc.x * loc.x + loc.y * loc.y);
    float s = n * n;
    return make3(lt.x * (-r * loc.y / s) + lt.y * r * loc.x / s, lt.z,
                 lt.x * loc.x / n + lt.y * loc.y / n);
}

/* returns false when the sample set is rejected (reference returns nullptr) */
TM_HD bool traits_init_from_samples(int kind, const float* model_state, const float* smp, float* st) {
    float threshold = model_state[1];
    if (kind == 0) { /* identity_traits.hpp:23-29 */
        frame_store(make3(1, 0, 0), make3(0, 1, 0), make3(0, 0, 1), make3(0, 0, 0), 0.f, 0.f, st);
        return true;
    }
    if (kind == 1) { /* plane_traits.hpp:44-61 */
        f3 n = make3(smp[4], smp[5], smp[6]);
        f3 r1 = unit_orthogonal(n);
        f3 r0 = unit(cross3(r1, n));
        frame_store(r0, r1, n, make3(smp[0], smp[1], smp[2]), 0.f, threshold, st);
        return true;
    }
    if (kind == 2) { /* plane2_traits.hpp:48-84 */
        f3 s1 = make3(smp[0], smp[1], smp[2]);
        f3 s2 = make3(smp[12], smp[13], smp[14]);
        f3 s3 = make3(smp[24], smp[25], smp[26]);
        f3 d1 = s2 - s1, d2 = s3 - s1;
        f3 nrm = unit(cross3(d1, d2));
        float c1 = dot3(make3(smp[4], smp[5], smp[6]), nrm);
        float c2 = dot3(make3(smp[16], smp[17], smp[18]), nrm);
        float c3 = dot3(make3(smp[28], smp[29], smp[30]), nrm);
        if (1.f - fabsf(c1) > 0.1f) return false;
        if (1.f - fabsf(c2) > 0.1f) return false;
        if (1.f - fabsf(c3) > 0.1f) return false;
        d1 = d1 - scale(nrm, dot3(d1, nrm));
        d1 = unit(d1);
        d2 = unit(cross3(nrm, d1));
        frame_store(d1, d2, nrm, s1, 0.f, threshold, st);
        return true;
    }
    /* cylinder_traits.hpp:53-97: closest approach of the two normal lines */
    f3 p1 = make3(smp[0], smp[1], smp[2]), p2 = make3(smp[12], smp[13], smp[14]);
    f3 n1 = make3(smp[4], smp[5], smp[6]), n2 = make3(smp[16], smp[17], smp[18]);
    f3 w = (n1 + p1) - p2;
    float a = dot3(n1, n1), b = dot3(n1, n2), c = dot3(n2, n2), d = dot3(n1, w), e = dot3(n2, w);
    float den = a * c - b * b;
    float sc, tc;
    if ((double)den < 1e-8) {
        sc = 0.f;
        tc = (b > c ? d / b : e / c);
    } else {
        sc = (b * e - c * d) / den;
        tc = (a * e - b * d) / den;
    }
    f3 origin = (p1 + n1) + scale(n1, sc);
    f3 axis = unit((p2 + scale(n2, tc)) - origin);
    f3 ref = unit_orthogonal(axis);
    float radius = len(cross3(axis, origin - p1));
    frame_store(ref, cross3(ref, axis), axis, origin, radius, threshold, st);
    return true;
}

/* project(): xyz -> uvw; false = nullopt */
TM_HD bool traits_project(int kind, const frame_t& f, f3 xyz, f3& uvw) {
    if (kind == 0) {
        uvw = xyz;
        return true;
    }
    f3 loc = frame_point(f, xyz);
    if (kind == 1) { /* plane_traits.hpp:64-72 */
        if (fabsf(loc.z) > f.threshold) return false;
        uvw = loc;
        return true;
    }
    if (kind == 2) { /* plane2_traits.hpp:86-88 */
        uvw = loc;
        return true;
    }
    /* cylinder_traits.hpp:100-114 */
    float height = sqrtf(loc.x * loc.x + loc.y * loc.y) - f.radius;
    if (fabsf(height) > f.threshold) return false;
    float u = atan2_def(loc.y, loc.x);
    if (u < 0.f) u = (float)((double)u + 6.283185307179586);
    uvw = make3(u * f.radius, loc.z, height / f.radius);
    return true;
}

TM_HD f3 traits_tangent(int kind, const frame_t& f, f3 pos, f3 tgt) {
    if (kind == 0) return tgt;
    f3 lt = frame_dir(f, tgt);
    if (kind == 1) return lt;       /* plane_traits.hpp:81-85 */
    if (kind == 2) return unit(lt); /* plane2_traits.hpp:96-102 */
    f3 loc = frame_point(f, pos);   /* cylinder_traits.hpp:134-153 */
    return unit(cylinder_jacobian(f.radius, loc, unit(lt)));
}

TM_HD f3 traits_normal(int kind, const frame_t& f, f3 pos, f3 nrm, f3 tgt) {
    if (kind == 0) return nrm;
    if (kind == 1 || kind == 2) return unit(frame_dir(f, nrm)); /* plane*_traits.hpp normal() */
    /* cylinder_traits.hpp:156-178 */
    f3 g2 = unit(cross3(tgt, nrm));
    f3 loc = frame_point(f, pos);
    f3 l1 = unit(frame_dir(f, tgt));
    f3 l2 = unit(frame_dir(f, g2));
    f3 t1 = cylinder_jacobian(f.radius, loc, l1);
    f3 t2 = cylinder_jacobian(f.radius, loc, l2);
    return unit(cross3(t1, t2));
}

/* the argument of traits_distance's square root for the projector metrics (kind 2, 3): same operations */
TM_HD float traits_distance2(int kind, float radius, f3 a, f3 b) {
    if (kind == 2) {
        float dx = b.x - a.x, dy = b.y - a.y;
        return dx * dx + dy * dy;
    }
    float du = fabsf(b.x - a.x), dv = fabsf(b.y - a.y);
    du = fminf(du, (float)(6.283185307179586 * (double)radius) - du);
    return du * du + dv * dv;
}

TM_HD float traits_distance(int kind, float radius, f3 a, f3 b) {
    if (kind == 0 || kind == 1) return len(b - a);
    if (kind == 2) { /* plane2_traits.hpp:112-116 */
        float dx = b.x - a.x, dy = b.y - a.y;
        return sqrtf(dx * dx + dy * dy);
    }
    /* cylinder_traits.hpp:181-189: wrap-around in u */
    float du = fabsf(b.x - a.x), dv = fabsf(b.y - a.y);
    du = fminf(du, (float)(6.283185307179586 * (double)radius) - du);
    return sqrtf(du * du + dv * dv);
}

}  // namespace tmx

#endif /* TM_TRAITS_MATH_H_ */
