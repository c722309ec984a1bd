"""The reference's search loops written a second time, in Python/numpy, from the reference TEXT.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ and by
tests/golden/make_search_traces.py. Nothing under triplet_match_amd/ or include/ may use it.

What is restated here (all citations into /root/reference):
  * scene<P>::impl::find_all_parallel   include/impl/scene.hpp:39-98
  * scene<P>::impl::find_parallel       include/impl/scene.hpp:108-240
  * scene<P>::impl::find_in_subset      include/impl/scene.hpp:242-367
  * finish_find :100-106, icp_ :369-404, project_ :411-510, base_transform_ :538-567
  * model<P>::impl::init / query / voxel_query   include/impl/model.hpp:15-192
  * feature / discretize_feature / valid / valid_bounds   include/impl/feature.hpp, src/discretize.cpp:20-30

It shares no code with include/triplet_match/search.hpp (C++) nor with oracle/tm_oracle.cpp (C++): it is a
third, separately written statement of the same text, in another language, used to pin the product's sampling
loops through traces committed under tests/golden/.

Where the reference leaves behaviour to a third party or to the clock, the SAME DEFINITIONS as the rest of the
project are taken (DESIGN.md "Numerics" / "defined stand-ins"):
  * RNG: std::mt19937 (the published MT19937 recurrence, written out below) seeded with 13, the reference's
    `deterministic` branch (:123, :245);
  * view::sample(left, n) | action::shuffle == a partial Fisher-Yates shuffle whose draws are `rng() % (n - k)`;
    the shuffle of a neighbourhood (:285) is run lazily, position k drawn when the inner loop reaches it;
  * hardware_concurrency() - 1 (:146) is the parameter `workers`; chunks are searched one after another
    (each seeds its own generator, :244-250, and touches fp_mask_ only at its own first points);
  * radius_search_inclusive(r, p) == {q : |q - p|^2 <= r * r}, ascending index (PCL/FLANN order is unpinned);
  * nearest model point of a voxel centre: smallest float32 squared distance, lowest index on ties; the
    centre is inv(to_voxel) * (i, j, k, 1) with inv_s = 1 / s, inv_t = -(t * inv_s);
  * unordered_multimap::equal_range yields equal keys in REVERSE insertion order (libstdc++, probed: SURVEY Q7);
  * std::sort of the candidates (:193) is taken as stable;
  * float32 evaluation order: dot = (a0 b0 + a1 b1) + a2 b2, 4x4 * (x, y, z, 1) row = ((m0 x + m4 y) + m8 z) + m12,
    Eigen's fixed 3x3 inverse = cofactors * (1 / det), normalize() = v / sqrt(squaredNorm) when squaredNorm > 0;
  * atan2f: the project's own (Cephes atanf reduction + degree-4 odd polynomial), restated below from the
    published algorithm; the reference's binary calls glibc (census: tests/golden/atan2_census.json);
  * log(1 - n_model / |left|) with |left| <= n_model (:133-134) is a NaN / -inf cast to uint64_t in the reference
    (undefined); defined as 0, so that min_sample_count applies; sample size is also capped by |left|;
  * Eigen::umeyama (:393): numpy float64 SVD form (no scale), rounded to float32.
"""
import hashlib
import math

import numpy as np

f32 = np.float32
MATCH_PROBABILITY = 0.999   # :14
MIN_SAMPLE_COUNT = 10       # :15
QUERY_LIMIT = 200           # :19
CURVATURE_K = 30            # :23


# ------------------------------------------------------------------------------------------------------------
# std::mt19937 (Matsumoto & Nishimura 1998, as mandated by [rand.predef]: 10000th draw of seed 5489 = 4123659995)
# ------------------------------------------------------------------------------------------------------------
class MT19937:
    N, M = 624, 397

    def __init__(self, seed):
        s = [0] * self.N
        s[0] = seed & 0xFFFFFFFF
        for i in range(1, self.N):
            s[i] = (1812433253 * (s[i - 1] ^ (s[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.s = s
        self.i = self.N
        self.draws = 0

    def _twist(self):
        s, N, M = self.s, self.N, self.M
        for k in range(N):
            y = (s[k] & 0x80000000) | (s[(k + 1) % N] & 0x7FFFFFFF)
            v = s[(k + M) % N] ^ (y >> 1)
            if y & 1:
                v ^= 0x9908B0DF
            s[k] = v
        self.i = 0

    def __call__(self):
        if self.i >= self.N:
            self._twist()
        y = self.s[self.i]
        self.i += 1
        self.draws += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def fnv1a64(arr):
    """FNV-1a over the little-endian bytes of an array (the fingerprint the C++ trace writer uses too)"""
    h = 1469598103934665603
    for b in np.ascontiguousarray(arr).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def bits(a):
    """float32 array -> list of its bit patterns (exact comparison through JSON)"""
    return [int(v) for v in np.ascontiguousarray(a, f32).ravel().view(np.uint32)]


# ------------------------------------------------------------------------------------------------------------
# float32 helpers in the project's defined association
# ------------------------------------------------------------------------------------------------------------
def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _sqn(a):
    return (a[..., 0] * a[..., 0] + a[..., 1] * a[..., 1]) + a[..., 2] * a[..., 2]


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def _normalized(a):
    z = _sqn(a)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where((z > 0)[..., None], a / np.sqrt(z)[..., None], a).astype(f32)


def _atanf(xx):
    """Cephes atanf (S. Moshier, public domain): range reduction at tan(3 pi / 8), tan(pi / 8) + odd polynomial"""
    xx = np.asarray(xx, f32)
    sign = np.where(xx < 0, f32(-1), f32(1)).astype(f32)
    x = np.abs(xx)
    big, mid = x > f32(2.414213562373095), x > f32(0.4142135623730950)
    with np.errstate(divide="ignore", invalid="ignore"):
        y = np.where(big, f32(1.57079632679489661923), np.where(mid, f32(0.78539816339744830962), f32(0))).astype(f32)
        xr = np.where(big, -(f32(1) / x), np.where(mid, (x - f32(1)) / (x + f32(1)), x)).astype(f32)
    z = xr * xr
    p = (((f32(8.05374449538e-2) * z - f32(1.38776856032e-1)) * z + f32(1.99777106478e-1)) * z
         - f32(3.33329491539e-1)) * z * xr + xr
    return (sign * (y + p)).astype(f32)


def atan2f(y, x):
    y, x = np.asarray(y, f32), np.asarray(x, f32)
    pi, pio2 = f32(3.14159265358979323846), f32(1.57079632679489661923)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = _atanf(y / x)
    w = np.where(x < 0, np.where(y < 0, -pi, pi), f32(0)).astype(f32)
    zero_x = np.where(y > 0, pio2, np.where(y < 0, -pio2, f32(0))).astype(f32)
    return np.where(x == 0, zero_x, w + z).astype(f32)


def _angle(a, b):  # impl/feature.hpp:5-8
    c = _cross(a, b)
    return atan2f(np.sqrt(_sqn(c)), np.abs(_dot(a, b)))


def feature(P0, T0, P1, T1):
    """impl/feature.hpp:16-34 for arrays of pairs: (f0, f1, f2); f3 == f0"""
    d0 = (P1 - P0).astype(f32)
    return np.sqrt(_sqn(d0)), _angle(d0, T0), _angle(d0, T1)


def discretize_range(v, mn, rg, steps):  # src/discretize.cpp:20-26
    with np.errstate(divide="ignore", invalid="ignore"):
        nval = ((v - mn) / rg).astype(f32)
        scaled = (nval * f32(steps)).astype(f32)
        k = np.where(np.isfinite(scaled), scaled, 0).astype(np.int64).astype(np.uint32)
    return np.where(nval < 0, np.uint32(0), np.where(nval >= 1, np.uint32((steps - 1) & 0xFFFFFFFF), k)).astype(np.uint32)


def discretize_step(v, step):  # src/discretize.cpp:28-30
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (v / step).astype(f32)
    return np.where(np.isfinite(q) & (q >= 0), q, 0).astype(np.int64).astype(np.uint32)


def base_transform(src_i, src_j, src_t, tgt_i, tgt_j, tgt_t):
    """impl/scene.hpp:538-567 for arrays (H, 3) of target triples and one source triple; returns (H, 16) col-major"""
    o_a = np.asarray(src_i, f32)
    u_a = _normalized((np.asarray(src_j, f32) - o_a).astype(f32))
    v_a = np.asarray(src_t, f32)
    v_a = _normalized((v_a - u_a * _dot(v_a, u_a)).astype(f32))
    w_a = _normalized(_cross(u_a, v_a))
    o_b = np.asarray(tgt_i, f32)
    u_b = _normalized((np.asarray(tgt_j, f32) - o_b).astype(f32))
    v_b = np.asarray(tgt_t, f32)
    v_b = _normalized((v_b - u_b * _dot(v_b, u_b)[..., None]).astype(f32))
    w_b = _normalized(_cross(u_b, v_b))
    A = np.stack([u_a, v_a, w_a], 1)  # A[r, c]: columns u, v, w (:557-558)

    def cof(i, j):
        i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
        return A[i1, j1] * A[i2, j2] - A[i1, j2] * A[i2, j1]
    det = (cof(0, 0) * A[0, 0] + cof(1, 0) * A[1, 0]) + cof(2, 0) * A[2, 0]
    invdet = f32(1) / det
    Ai = np.array([[cof(c, r) * invdet for c in range(3)] for r in range(3)], f32)  # base_a.inverse() (:563)
    B = np.stack([u_b, v_b, w_b], -1)  # (H, 3 rows, 3 cols)
    H = B.shape[0]
    T = np.zeros((H, 16), f32)
    R = np.zeros((H, 3, 3), f32)
    for r in range(3):
        for c in range(3):
            R[:, r, c] = (B[:, r, 0] * Ai[0, c] + B[:, r, 1] * Ai[1, c]) + B[:, r, 2] * Ai[2, c]
            T[:, c * 4 + r] = R[:, r, c]
    for r in range(3):
        T[:, 12 + r] = o_b[:, r] - ((R[:, r, 0] * o_a[0] + R[:, r, 1] * o_a[1]) + R[:, r, 2] * o_a[2])  # :564
    T[:, 15] = 1
    return T


# ------------------------------------------------------------------------------------------------------------
# model<P>::impl   (include/impl/model.hpp)
# ------------------------------------------------------------------------------------------------------------
class Model:
    def __init__(self, surfels, resolution, curvature, distance_step_count=20.0, angle_step=None,
                 min_diameter_factor=0.2, max_diameter_factor=1.0, voxel_chunk=2048):
        """init(sample_params) with an empty subset (:17-19); `curvature` = cloud->curvature(30, i) rows
        (pc_max_dir(3), pc_min, pc_max) — row f1 of SURVEY section 8, pinned separately against scipy/numpy"""
        S = np.ascontiguousarray(surfels, f32).reshape(-1, 12)
        self.cloud = S
        self.P, self.Nrm, self.Tg = S[:, 0:3], S[:, 4:7], S[:, 9:12]
        self.n = S.shape[0]
        self.resolution = f32(resolution)
        self.distance_step_count = f32(distance_step_count)
        self.angle_step = f32(angle_step) if angle_step is not None else f32(10.0) * f32(np.pi) / f32(180.0)
        finite = np.isfinite(self.P).all(1) & np.isfinite(self.Nrm).all(1) & np.isfinite(self.Tg).all(1)  # :24-30
        subset = np.nonzero(finite)[0]
        lo, hi = self.P[subset].min(0), self.P[subset].max(0)  # :34-38
        rg = (hi - lo).astype(f32)
        self.diameter = f32(np.sqrt(_sqn(rg)))  # :39
        cell = f32(0.5) * self.resolution
        ext = np.maximum(rg / cell, f32(1)).astype(f32)  # :45-46
        self.margin = 5
        self.extents = (ext + f32(2.0) * f32(self.margin)).astype(f32).astype(np.int32)  # :50 cast<int>
        scale = np.where(rg < f32(1e-5), f32(1), ext / np.where(rg == 0, f32(1), rg)).astype(f32)  # :52-55
        trans = ((scale * (-lo) + f32(self.margin)).astype(f32) - f32(0.5)).astype(f32)  # :58-61
        self.to_voxel = np.zeros(16, f32)
        self.to_voxel[[0, 5, 10]] = scale
        self.to_voxel[12:15] = trans
        self.to_voxel[15] = 1
        self._build_voxels(scale, trans, voxel_chunk)
        # :96-99
        tn = np.sqrt(_sqn(self.Tg[subset]))
        cv = np.asarray(curvature, f32).reshape(-1, 5)[subset]
        with np.errstate(divide="ignore", invalid="ignore"):
            keep = (tn > f32(0.7)) & ((cv[:, 3] / cv[:, 4]) < f32(0.2))
        self.subset = subset[keep].astype(np.uint32)
        self._build_pairs(f32(min_diameter_factor), f32(max_diameter_factor))

    def _build_voxels(self, scale, trans, chunk):
        """:63-94: voxel_data_[k * ex * ey + j * ex + i] = nearest point of the WHOLE cloud to inv * (i, j, k, 1)"""
        ex, ey, ez = (int(e) for e in self.extents)
        inv_s = (f32(1) / scale).astype(f32)
        inv_t = (-(trans * inv_s)).astype(f32)
        cx = (inv_s[0] * np.arange(ex, dtype=f32) + inv_t[0]).astype(f32)
        cy = (inv_s[1] * np.arange(ey, dtype=f32) + inv_t[1]).astype(f32)
        cz = (inv_s[2] * np.arange(ez, dtype=f32) + inv_t[2]).astype(f32)
        P = np.where(np.isfinite(self.P), self.P, f32(np.inf))  # a non-finite point is nobody's nearest
        px, py, pz = P[:, 0][None, :], P[:, 1][None, :], P[:, 2][None, :]
        total = ex * ey * ez
        vox = np.empty(total, np.uint32)
        lin = np.arange(total, dtype=np.int64)
        for c0 in range(0, total, chunk):
            l = lin[c0:c0 + chunk]
            i, j, k = l % ex, (l // ex) % ey, l // (ex * ey)
            dx, dy, dz = cx[i][:, None] - px, cy[j][:, None] - py, cz[k][:, None] - pz
            d2 = (dx * dx + dy * dy) + dz * dz
            vox[c0:c0 + chunk] = np.argmin(d2, 1)  # first minimum = lowest index
        self.voxels = vox

    def _pair_filters(self, I, J, lower, upper):
        d1 = (self.P[J] - self.P[I]).astype(f32)  # :107
        dist1 = np.sqrt(_sqn(d1))                 # :108
        with np.errstate(invalid="ignore", divide="ignore"):
            dn = (d1 / dist1[:, None]).astype(f32)  # :109
            ok = ~((dist1 < lower) | (dist1 > upper))  # :110
            ok &= ~((f32(1) - np.abs(_dot(dn, self.Tg[I]))) < f32(0.01))  # :111
        return ok

    def _build_pairs(self, min_f, max_f):
        sub = self.subset.astype(np.int64)
        I, J = np.meshgrid(sub, sub, indexing="ij")  # cartesian_product(subset_, subset_): i-major (:100)
        I, J = I.ravel(), J.ravel()
        ne = I != J  # :105
        I, J = I[ne], J[ne]
        lower, upper = self.diameter * min_f, self.diameter * max_f  # :101-102
        ok = self._pair_filters(I, J, lower, upper)
        I, J = I[ok], J[ok]
        f0, f1, f2 = feature(self.P[I], self.Tg[I], self.P[J], self.Tg[J])  # :113
        if len(f0) == 0:
            raise RuntimeError("no valid model pair")  # the reference dereferences end() at :151
        bmin = np.array([f0.min(), f1.min(), f2.min(), f0.min()], f32)  # :115 feat_bounds_.extend
        bmax = np.array([f0.max(), f1.max(), f2.max(), f0.max()], f32)
        diag = (bmax - bmin).astype(f32)
        # valid_bounds(bounds, pi / 10, 2 pi, 0.0f, 1.f) (:122 -> impl/feature.hpp:91-114): dims 0 and 3 only
        nmin, nmax = bmin.copy(), bmax.copy()
        for d in (0, 3):
            nmin[d] = bmin[d] + f32(0.0) * diag[d]
            nmax[d] = bmin[d] + f32(1.0) * diag[d]
        self.bmin, self.bmax = nmin, nmax
        ok = self.valid(f0, f1, f2)  # :135
        I, J, f0, f1, f2 = I[ok], J[ok], f0[ok], f1[ok], f2[ok]
        keys = self.discretize_feature(f0, f1, f2)  # :136
        self.map = {}
        for key, i, j in zip(map(tuple, keys.tolist()), I.tolist(), J.tolist()):  # :145 insert, in pair order
            self.map.setdefault(key, []).append((i, j))
        self.pair_count = int(len(I))

    def valid(self, f0, f1, f2):  # impl/feature.hpp:50-88
        pi = f32(np.pi)
        return ~((f0 < self.bmin[0]) | (f0 > self.bmax[0])) & (f1 >= 0) & (f1 <= pi) & (f2 >= 0) & (f2 <= pi)

    def discretize_feature(self, f0, f1, f2):  # impl/feature.hpp:38-46; float step count -> uint32_t steps (Q3)
        steps = int(self.distance_step_count)
        rg0 = f32(self.bmax[0] - self.bmin[0])  # bounds.diagonal()[0]
        k0 = discretize_range(np.asarray(f0, f32), self.bmin[0], rg0, steps)
        return np.stack([k0, discretize_step(np.asarray(f1, f32), self.angle_step),
                         discretize_step(np.asarray(f2, f32), self.angle_step), k0], -1)

    def query(self, key):
        """:169-178 + the caller's truncation (impl/scene.hpp:307-310): first query_limit entries of equal_range,
        libstdc++ order = reverse insertion"""
        e = self.map.get(tuple(int(k) for k in key), [])
        return e[::-1][:QUERY_LIMIT]

    def table_fingerprint(self):
        """all entries, keys ascending, container order inside a key"""
        out = []
        for key in sorted(self.map):
            for (i, j) in self.map[key][::-1]:
                out.append((*key, i, j))
        return fnv1a64(np.array(out, np.uint32))


# ------------------------------------------------------------------------------------------------------------
# scene<P>::impl   (include/impl/scene.hpp)
# ------------------------------------------------------------------------------------------------------------
class Scene:
    def __init__(self, surfels, curvature=None, tangent_mask=None):
        S = np.ascontiguousarray(surfels, f32).reshape(-1, 12)
        self.cloud = S
        self.P, self.Nrm, self.Tg = S[:, 0:3], S[:, 4:7], S[:, 9:12]
        self.n = S.shape[0]
        if tangent_mask is None:  # :46-54
            cv = np.asarray(curvature, f32).reshape(-1, 5)
            with np.errstate(divide="ignore", invalid="ignore"):
                tangent_mask = (np.sqrt(_sqn(self.Tg)) > f32(0.7)) & ((cv[:, 3] / cv[:, 4]) < f32(0.2))
        self.tangent_mask = np.asarray(tangent_mask).astype(bool)
        self.mask = np.zeros(self.n, bool)
        self.fp_mask = np.zeros(self.n, bool)
        self.trace = None

    # ---- project_ (:411-510) for a batch of poses over one subset -------------------------------------------
    def _outcomes(self, m, poses, subset, dist_thres):
        """per (pose, subset position): reaches the end of the loop body (not masked, inside the grid: :434, :458),
        is a correspondence (:465-488), model partner, float32 term |ref . ref_n| (:483)"""
        T = np.asarray(poses, f32).reshape(-1, 16)
        idx = np.asarray(subset, np.int64)
        x, y, z = (self.P[idx, c][None, :] for c in range(3))

        def row(r):  # :444
            return ((T[:, r, None] * x + T[:, 4 + r, None] * y) + T[:, 8 + r, None] * z) + T[:, 12 + r, None]
        with np.errstate(invalid="ignore", over="ignore"):
            px, py, pz = row(0), row(1), row(2)
            V = m.to_voxel
            v = [V[0] * px + V[12], V[5] * py + V[13], V[10] * pz + V[14]]  # impl/model.hpp:182
            fin = np.isfinite(v[0]) & np.isfinite(v[1]) & np.isfinite(v[2])
            ijk = [np.where(fin, np.trunc(np.where(fin, c, 0)), -1).astype(np.int64) for c in v]  # cast<int>
        ex, ey, ez = (int(e) for e in m.extents)
        inside = ((ijk[0] >= 0) & (ijk[0] < ex) & (ijk[1] >= 0) & (ijk[1] < ey) & (ijk[2] >= 0) & (ijk[2] < ez))  # :186
        lin = np.where(inside, (ijk[2] * ey + ijk[1]) * ex + ijk[0], 0)
        pm = m.voxels[lin].astype(np.int64)  # :190-191
        with np.errstate(invalid="ignore", over="ignore"):
            dx, dy, dz = px - m.P[pm, 0], py - m.P[pm, 1], pz - m.P[pm, 2]
            dist = np.sqrt((dx * dx + dy * dy) + dz * dz)  # :464
        thres = f32(dist_thres) * m.resolution  # :413
        reach = inside & ~self.mask[idx][None, :]
        close = reach & ~(dist > thres)  # :465
        m_is_t = np.sqrt(_sqn(m.Tg))[pm] > f32(0.7)  # :469-470
        s_is_t = self.tangent_mask[idx][None, :]     # :441
        corr = close & (m_is_t == s_is_t)            # :472-478
        sv = np.where(self.tangent_mask[idx][:, None], self.Tg[idx], self.Nrm[idx]).astype(f32)  # :442
        mref = np.where((np.sqrt(_sqn(m.Tg)) > f32(0.7))[:, None], m.Tg, m.Nrm).astype(f32)       # :469, :479-481
        with np.errstate(invalid="ignore", over="ignore"):
            rx = (T[:, 0, None] * sv[None, :, 0] + T[:, 4, None] * sv[None, :, 1]) + T[:, 8, None] * sv[None, :, 2]  # :461
            ry = (T[:, 1, None] * sv[None, :, 0] + T[:, 5, None] * sv[None, :, 1]) + T[:, 9, None] * sv[None, :, 2]
            rz = (T[:, 2, None] * sv[None, :, 0] + T[:, 6, None] * sv[None, :, 1]) + T[:, 10, None] * sv[None, :, 2]
            term = np.abs((rx * mref[pm, 0] + ry * mref[pm, 1]) + rz * mref[pm, 2])
        return reach, corr, pm, np.where(corr, term, f32(0)).astype(f32)

    @staticmethod
    def _early_drop_upper(tried, S, corrs):
        """:493-498, the literal expression; casts of negative doubles wrap as x86-64 cvttsd2si does"""
        N = -2.0 - tried
        x = -2.0 - S
        n = -1.0 - corrs
        tmp = math.sqrt((x * n * (N - x) * (N - n)) / (N - 1.0))
        inner = int((x * n + tmp) / N) % (1 << 32)
        return int(-1.0 - inner) % (1 << 32)

    def project_batch(self, m, poses, subset, accept_prob, dist_thres, early_out):
        """project_ for every pose: list of (n_corrs, score, saved). With early_out the drop rule of :492-506 is
        applied: the test runs at the END of a body that reached it, for the first position with tried >= tests[k]."""
        S = len(subset)
        n_model = m.n
        reach, corr, pm, term = self._outcomes(m, poses, subset, dist_thres)
        ccum = np.cumsum(corr, 1)
        scum = np.cumsum(term.astype(np.float64), 1)  # sequential double sum in scan order (:429, :483)
        out = []
        tests = [int(f32(f32(f32(0.05) * f32(i + 1)) * f32(S))) for i in range(int(f32(1.0) / f32(0.05)) - 2)]  # :422-426
        thr = f32(accept_prob) * f32(n_model)  # :500 float * size_t
        for h in range(reach.shape[0]):
            dropped = None
            if early_out and S:
                pos = np.nonzero(reach[h])[0] + 1  # values of `tried` at which the end of the body is reached
                t_prev, k = 0, 0
                while k < len(tests):
                    a = np.searchsorted(pos, max(tests[k], t_prev + 1))
                    if a >= len(pos):
                        break
                    tried = int(pos[a])
                    upper = self._early_drop_upper(tried, S, int(ccum[h, tried - 1]))
                    if f32(upper) < thr:  # uint32_t < float
                        dropped = tried
                        break
                    k += 1
                    t_prev = tried
            if dropped is not None:
                out.append((int(ccum[h, dropped - 1]), float(scum[h, dropped - 1]), S - dropped))  # :501-502 raw sum
            elif S:
                out.append((int(ccum[h, -1]), float(scum[h, -1]) / n_model, 0))  # :509
            else:
                out.append((0, 0.0, 0))
        return out

    def project_literal(self, m, pose, subset, accept_prob, dist_thres, early_out):
        """the same loop statement by statement (slow; cross-check of project_batch)"""
        reach, corr, pm, term = self._outcomes(m, np.asarray(pose, f32).reshape(1, 16), subset, dist_thres)
        S = len(subset)
        tests = [int(f32(f32(f32(0.05) * f32(i + 1)) * f32(S))) for i in range(int(f32(1.0) / f32(0.05)) - 2)]
        tried = next_test = n_c = 0
        score = 0.0
        thr = f32(accept_prob) * f32(m.n)
        for k in range(S):
            tried += 1
            if not reach[0, k]:
                continue
            if corr[0, k]:
                score += float(np.float64(term[0, k]))
                n_c += 1
            if early_out and next_test < len(tests) and tried >= tests[next_test]:
                if f32(self._early_drop_upper(tried, S, n_c)) < thr:
                    return n_c, score, S - tried
                next_test += 1
        return n_c, score / m.n, 0

    def finish_find(self, m, pose, dist_thres):
        """:100-106: project_ over all scene indices, no early drop -> (scene_corrs, model_corrs, score)"""
        sc, mc, tot = [], [], 0.0
        for c0 in range(0, self.n, 1 << 16):
            sub = np.arange(c0, min(self.n, c0 + (1 << 16)))
            reach, corr, pm, term = self._outcomes(m, np.asarray(pose, f32).reshape(1, 16), sub, dist_thres)
            sel = np.nonzero(corr[0])[0]
            sc.append(sub[sel])
            mc.append(pm[0, sel])
            for t in term[0, sel].astype(np.float64):  # sequential sum
                tot += float(t)
        return (np.concatenate(sc).astype(np.uint32), np.concatenate(mc).astype(np.uint32), tot / m.n)

    # ---- icp_ (:369-404) ---------------------------------------------------------------------------------------
    def icp(self, m, pose, max_iterations, dist_thres):
        if max_iterations == 0:
            return None
        best = self.finish_find(m, pose, 2 * dist_thres)  # :373
        best_t = np.asarray(pose, f32)
        it = 0
        while True:
            if len(best[0]) < 3:
                break
            scn = self.P[best[0].astype(np.int64)].astype(np.float64)
            mdl = m.P[best[1].astype(np.int64)].astype(np.float64)
            cs, cm = scn.mean(0), mdl.mean(0)
            sigma = (mdl - cm).T @ (scn - cs) / len(scn)  # Eigen::umeyama(scn, mdl, false), :393
            U, d, Vt = np.linalg.svd(sigma)
            Sg = np.eye(3)
            if np.linalg.det(U) * np.linalg.det(Vt) < 0:
                Sg[2, 2] = -1
            R = U @ Sg @ Vt
            t = cm - R @ cs
            T = np.zeros(16, f32)
            for r in range(3):
                for c in range(3):
                    T[c * 4 + r] = R[r, c]
                T[12 + r] = t[r]
            T[15] = 1
            proj = self.finish_find(m, T, 2 * dist_thres)  # :395
            if len(proj[0]) < len(best[0]):  # :396
                break
            best, best_t = proj, T
            it += 1
            if it == max_iterations:
                break
        return best_t, best

    # ---- find_in_subset (:242-367) ---------------------------------------------------------------------------
    def find_in_subset(self, m, firsts, lower, upper, dist_thres, accept_prob, early_out_threshold, force_up,
                       seed, early_drop, dead_work, sample_all, rec):
        rng = MT19937(seed)  # :244-250, deterministic branch
        best_score = 0.0
        best_ts = []
        out = False
        cons = 0
        r2 = m.diameter * m.diameter
        for i in firsts:
            if out:
                break  # :267
            i = int(i)
            if self.fp_mask[i] or self.mask[i]:
                continue  # :268
            cons += 1
            d = (self.P - self.P[i]).astype(f32)
            nn = np.nonzero(_sqn(d) <= r2)[0]  # :273 radius_search_inclusive(m.diameter(), p1)
            if len(nn) == 0:
                continue
            prob = float(m.n) / len(nn)  # :276
            inner_bound = int(-math.log(1.0 - MATCH_PROBABILITY) / prob)  # :277
            inner_bound = max(inner_bound, MIN_SAMPLE_COUNT)
            inner_bound = min(inner_bound, len(nn))
            first_rec = dict(i=i, nn=int(len(nn)), inner_bound=inner_bound, draws=0, js=[], hyps=0)
            rec["firsts"].append(first_rec)
            first_rec["dead"] = bool(best_ts)  # the chunk has its candidate: nothing below can change the result
            if best_ts and not dead_work and not sample_all:
                first_rec["sampled"] = False
                continue
            first_rec["sampled"] = True
            inner0 = nn.copy()  # :284
            n_nn = len(nn)
            valid_samples = 0
            for k in range(n_nn):  # :285 shuffle, run lazily: position k is drawn when the loop reaches it
                first_rec["draws"] += 1
                if k + 1 < n_nn:
                    r = k + rng() % (n_nn - k)
                    inner0[k], inner0[r] = inner0[r], inner0[k]
                j = int(inner0[k])
                if out:
                    break  # :289
                if not self.tangent_mask[j] or self.mask[j] or i == j:
                    continue  # :290
                d0 = (self.P[j] - self.P[i]).astype(f32)  # :293
                sqn0 = _sqn(d0)
                d0 = _normalized(d0)
                if sqn0 < lower or sqn0 > upper:
                    continue  # :296
                if f32(1) - np.abs(_dot(d0, self.Tg[i])) < f32(0.01):
                    continue  # :297
                f0, f1, f2 = feature(self.P[i][None], self.Tg[i][None], self.P[j][None], self.Tg[j][None])  # :299
                if not m.valid(f0, f1, f2)[0]:
                    continue  # :300
                entries = m.query(m.discretize_feature(f0, f1, f2)[0])  # :304
                valid_samples += 1
                first_rec["js"].append(j)
                if entries and (not best_ts or dead_work):
                    mi = np.array([e[0] for e in entries], np.int64)
                    mj = np.array([e[1] for e in entries], np.int64)
                    T = base_transform(self.P[i], self.P[j], self.Tg[i], m.P[mi], m.P[mj], m.Tg[mi])  # :315
                    keep = np.ones(len(T), bool)
                    if force_up:
                        keep = ~(np.abs(f32(1) - T[:, 10]) > f32(0.01))  # :317
                    T, mi, mj = T[keep], mi[keep], mj[keep]
                    first_rec["hyps"] += int(len(T))
                    if len(T):
                        res = self.project_batch(m, T, nn, accept_prob, dist_thres, early_drop)  # :326
                        for h, (n_c, score, saved) in enumerate(res):
                            if out:
                                break  # :309
                            if score > best_score:  # :330
                                if not best_ts and f32(n_c) > f32(accept_prob) * f32(m.n):  # :331 size_t > float
                                    best_score = score
                                    best_ts.insert(0, T[h].copy())  # :335-337
                                    rec["hit"] = dict(i=i, j=j, m_i=int(mi[h]), m_j=int(mj[h]), entry=int(h),
                                                      count=int(n_c), pose=bits(T[h]))
                                if best_score >= early_out_threshold:  # :341, detail::early_out
                                    out = True
                if valid_samples > inner_bound:
                    break  # :350
        rec["cons"] = cons
        matches = []
        for T in best_ts[:5]:  # :361-364
            matches.append((T, self.finish_find(m, T, dist_thres)))
        return matches, cons

    # ---- find_parallel (:108-240) ----------------------------------------------------------------------------
    def find_parallel(self, m, tangent_indices, n_scene, dist_thres, max_icp_iterations, accept_prob,
                      early_out_threshold, sp, workers, seed, early_drop, dead_work, sample_all, rrec):
        n_model = len(m.subset)  # m.point_count()
        n_model_all = m.n
        min_points = int(f32(f32(1.0) * f32(accept_prob)) * f32(n_model))  # :114
        min_points_all = int(f32(accept_prob) * f32(n_model_all))          # :115
        lower = m.diameter * f32(sp["min_diameter_factor"])
        upper = m.diameter * f32(sp["max_diameter_factor"])
        lower, upper = lower * lower, upper * upper  # :119-120
        rng = MT19937(seed)
        left = [int(i) for i in tangent_indices if not self.fp_mask[i] and not self.mask[i]]  # :131
        rrec.update(left=len(left), min_points=min_points, min_points_all=min_points_all)
        if not left:
            rrec.update(outer_bound=0, outer=[], chunks=[], candidates=[], accepted=[])
            return []
        outer_prob = n_model / len(left)  # :133
        if outer_prob >= 1.0:
            outer_bound = 0  # log of a non-positive number cast to uint64_t: undefined in the reference, defined as 0
        else:
            outer_bound = int(math.log(1.0 - MATCH_PROBABILITY) / math.log(1.0 - outer_prob))  # :134
        outer_bound = max(outer_bound, MIN_SAMPLE_COUNT)
        outer_bound = min(outer_bound, n_scene)  # :137
        outer_bound = min(outer_bound, len(left))  # view::sample yields at most |left|
        for k in range(outer_bound):  # :143-144 sample + shuffle
            if k + 1 < len(left):
                r = k + rng() % (len(left) - k)
                left[k], left[r] = left[r], left[k]
        outer = left[:outer_bound]
        concurrency = max(1, workers)  # :146
        batch = outer_bound // concurrency
        if outer_bound % concurrency > 0:
            batch += 1  # :147-150
        rrec.update(outer_bound=outer_bound, outer=list(outer), batch_size=batch, chunks=[])
        cands = []
        for c0 in range(0, len(outer), batch):  # :156-166
            chunk = outer[c0:c0 + batch]
            crec = dict(firsts=[], hit=None)
            rrec["chunks"].append(crec)
            matches, cons = self.find_in_subset(m, chunk, lower, upper, dist_thres, accept_prob, early_out_threshold,
                                                sp["force_up"], seed, early_drop, dead_work, sample_all, crec)
            for idx in chunk[:cons]:
                self.fp_mask[idx] = True  # :160-162 take_exactly(chunk, cons)
            for (T, ff) in matches:  # :169-171
                refined = self.icp(m, T, max_icp_iterations, dist_thres)
                if refined is None:
                    cands.append((T, ff[0], ff[1], ff[2]))
                else:
                    cands.append((refined[0], refined[1][0], refined[1][1], refined[1][2]))
        rrec["candidates"] = [dict(pose=bits(c[0]), n=int(len(c[1])), scene=fnv1a64(c[1]), model=fnv1a64(c[2]),
                                   score=float(c[3])) for c in cands]
        order = sorted(range(len(cands)), key=lambda k: -len(cands[k][1]))  # :193-195 (stable)
        result = []
        used = set()
        for k in order:  # :200-237
            T, is_, im_, sscore = cands[k]
            used_copy = set(used)
            isres, imres = [], []
            for a in range(len(is_)):
                if int(is_[a]) not in used:
                    isres.append(int(is_[a]))
                    imres.append(int(im_[a]))
                used.add(int(is_[a]))
            score = sum(1 for idx in isres if self.tangent_mask[idx])
            score_all = len(isres)
            if score < min_points or score_all < min_points_all:  # :227
                used = used_copy
                continue
            result.append((T, np.array(isres, np.uint32), np.array(imres, np.uint32), sscore))
        rrec["accepted"] = [dict(pose=bits(c[0]), n=int(len(c[1])), scene=fnv1a64(c[1]), model=fnv1a64(c[2]),
                                 score=float(c[3])) for c in result]
        return result

    # ---- find_all_parallel (:39-98) --------------------------------------------------------------------------
    def find_all_parallel(self, m, dist_thres, model_match_factor, early_out_factor, sample_params,
                          max_icp_iterations, workers=7, seed=13, early_drop=True, dead_work=False, sample_all=True,
                          max_rounds=64):
        """returns (matches with the scene->model transform (before the inversion at :92), trace).
        Once a chunk has its candidate the reference keeps sampling and keeps calling project_, but nothing of it can
        change the result any more: best_ts.empty() gates :331 and best_score, hence `out`, is written only there.
        dead_work=True runs those calls too; sample_all=True (default) runs the sampling of the remaining first points
        without their project_ calls (their trace entries carry `dead: true`); with both off they are skipped."""
        self.mask[:] = False
        self.fp_mask[:] = False
        tangent_indices = np.nonzero(self.tangent_mask)[0]  # :48-54
        n_scene = len(tangent_indices)
        possible_model = m.n  # :63-68 considered_correspondence_ is true for every point (samples_on_all)
        early_out_threshold = int(f32(early_out_factor) * f32(possible_model))  # :78 float -> uint32_t parameter
        trace = dict(rounds=[])
        results = []
        for _ in range(max_rounds):
            rrec = {}
            trace["rounds"].append(rrec)
            matches = self.find_parallel(m, tangent_indices, n_scene, dist_thres, max_icp_iterations, model_match_factor,
                                         early_out_threshold, sample_params, workers, seed, early_drop, dead_work,
                                         sample_all, rrec)
            if not matches:
                break  # :81-83
            for (T, sc, mc, score) in matches:
                self.mask[sc.astype(np.int64)] = True  # :88-90
                results.append((T, sc, mc, score))
        return results, trace


def sha256_of(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
