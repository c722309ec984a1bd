"""ctypes binding of the CPU ORACLE (oracle/libtm_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. Nothing under triplet_match_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtm_oracle.so")

u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "tm_oracle.cpp")
    hdr = os.path.join(_HERE, "tm_oracle.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        if not os.path.exists(src):
            raise RuntimeError("oracle sources missing")
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtm_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.tmo_murmur3_32.restype = C.c_uint32
        L.tmo_murmur3_32.argtypes = [u32p, C.c_int, C.c_uint32]
        L.tmo_murmur_key.restype = C.c_uint32
        L.tmo_murmur_key.argtypes = [u32p, C.c_int]
        L.tmo_discretize_range.restype = C.c_uint32
        L.tmo_discretize_range.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint32]
        L.tmo_discretize_step.restype = C.c_uint32
        L.tmo_discretize_step.argtypes = [C.c_float, C.c_float]
        L.tmo_atan2f.restype = C.c_float
        L.tmo_atan2f.argtypes = [C.c_float, C.c_float]
        L.tmo_feature.argtypes = [f32p, f32p, f32p]
        L.tmo_discretize_feature.argtypes = [f32p, f32p, f32p, C.c_float, C.c_float, u32p]
        L.tmo_valid.restype = C.c_int
        L.tmo_valid.argtypes = [f32p, f32p, f32p]
        L.tmo_valid_bounds.argtypes = [f32p, f32p, C.c_float, C.c_float, f32p, f32p]
        L.tmo_base_transform.argtypes = [f32p] * 7
        L.tmo_early_drop_upper.restype = C.c_uint32
        L.tmo_early_drop_upper.argtypes = [C.c_uint32] * 3
        L.tmo_early_drop_upper_literal.restype = C.c_uint32
        L.tmo_early_drop_upper_literal.argtypes = [C.c_uint32] * 3
        L.tmo_early_drop_checkpoints.argtypes = [C.c_uint32, u32p]
        L.tmo_atan2_census.restype = C.c_uint64
        L.tmo_atan2_census.argtypes = [f32p, C.c_uint32, u32p, C.c_uint32, u32p, C.c_uint64, C.c_float, C.c_float, C.c_float,
                                       u32p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), f32p]
        L.tmo_early_drop_compare.restype = C.c_uint64
        L.tmo_early_drop_compare.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), u32p,
                                             C.c_uint32]
        L.tmo_cloud_resolution.restype = C.c_float
        L.tmo_cloud_resolution.argtypes = [f32p, C.c_uint32, f32p, C.c_int]
        L.tmo_cloud_curvature.argtypes = [f32p, C.c_uint32, C.c_uint32, f32p, C.c_int]
        L.tmo_cloud_knn.restype = C.c_uint32
        L.tmo_cloud_knn.argtypes = [f32p, C.c_uint32, u32p, C.c_uint32, f32p, C.c_uint32, i32p, f32p]
        L.tmo_cloud_radius_search.restype = C.c_uint32
        L.tmo_cloud_radius_search.argtypes = [f32p, C.c_uint32, u32p, C.c_uint32, f32p, C.c_float, i32p, f32p,
                                              C.c_uint32]
        L.tmo_cloud_curvature_of.argtypes = [f32p, C.c_uint32, i32p, C.c_uint32, f32p]
        L.tmo_model_create.restype = C.c_void_p
        L.tmo_model_create.argtypes = [f32p, C.c_uint32, C.c_float, C.c_float, C.c_float]
        L.tmo_model_destroy.argtypes = [C.c_void_p]
        L.tmo_model_set_grid_subset.restype = C.c_int
        L.tmo_model_set_grid_subset.argtypes = [C.c_void_p, u32p, C.c_uint32]
        L.tmo_model_build_voxels.restype = C.c_int
        L.tmo_model_build_voxels.argtypes = [C.c_void_p, C.c_int]
        L.tmo_model_build_voxels_bruteforce.restype = C.c_int
        L.tmo_model_build_voxels_bruteforce.argtypes = [C.c_void_p, C.c_int]
        L.tmo_model_set_voxels.restype = C.c_int
        L.tmo_model_set_voxels.argtypes = [C.c_void_p, u32p, C.c_uint64]
        L.tmo_model_build_pairs.restype = C.c_int
        L.tmo_model_build_pairs.argtypes = [C.c_void_p, u32p, C.c_uint32, C.c_float, C.c_float]
        L.tmo_model_diameter.restype = C.c_float
        L.tmo_model_diameter.argtypes = [C.c_void_p]
        L.tmo_model_set_metric.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.tmo_model_size.restype = C.c_uint32
        L.tmo_model_size.argtypes = [C.c_void_p]
        L.tmo_model_point_count.restype = C.c_uint32
        L.tmo_model_point_count.argtypes = [C.c_void_p]
        L.tmo_model_get_grid.argtypes = [C.c_void_p, i32p, f32p]
        L.tmo_model_voxel_count.restype = C.c_uint64
        L.tmo_model_voxel_count.argtypes = [C.c_void_p]
        L.tmo_model_voxels.restype = u32p
        L.tmo_model_voxels.argtypes = [C.c_void_p]
        L.tmo_model_get_feature_bounds.argtypes = [C.c_void_p, f32p, f32p]
        L.tmo_model_pair_count.restype = C.c_uint64
        L.tmo_model_pair_count.argtypes = [C.c_void_p]
        L.tmo_model_valid_count.restype = C.c_uint64
        L.tmo_model_valid_count.argtypes = [C.c_void_p]
        L.tmo_model_query.restype = C.c_int
        L.tmo_model_query.argtypes = [C.c_void_p, f32p, u32p, C.c_uint32, u32p]
        L.tmo_model_query_key.restype = C.c_int
        L.tmo_model_query_key.argtypes = [C.c_void_p, u32p, u32p, C.c_uint32, u32p]
        L.tmo_model_table_keys.restype = C.c_uint32
        L.tmo_model_table_keys.argtypes = [C.c_void_p, u32p, u32p, C.c_uint32]
        L.tmo_model_voxel_query.restype = C.c_int
        L.tmo_model_voxel_query.argtypes = [C.c_void_p, f32p, u32p]
        L.tmo_scene_create.restype = C.c_void_p
        L.tmo_scene_create.argtypes = [f32p, C.c_uint32, i32p]
        L.tmo_scene_destroy.argtypes = [C.c_void_p]
        L.tmo_scene_set_mask.argtypes = [C.c_void_p, i32p]
        L.tmo_project.restype = C.c_uint32
        L.tmo_project.argtypes = [C.c_void_p, C.c_void_p, i32p, C.c_uint32, f32p, C.c_float,
                                  C.c_float, C.c_int, u32p, u32p, f64p, u32p]
        L.tmo_generate_hypotheses.restype = C.c_int
        L.tmo_generate_hypotheses.argtypes = [C.c_void_p, C.c_void_p, u32p, C.c_uint32, C.c_float,
                                              C.c_float, C.c_int, C.c_int, f32p, u32p, C.c_uint32,
                                              u32p]
        L.tmo_score_hypotheses.argtypes = [C.c_void_p, C.c_void_p, f32p, C.c_uint32, i32p,
                                           C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int, u32p,
                                           f64p, u64p]
        L.tmo_scene_radius_subset.restype = C.c_uint32
        L.tmo_scene_radius_subset.argtypes = [C.c_void_p, C.c_uint32, C.c_float, i32p, C.c_uint32]
        L.tmo_traits_init_from_samples.restype = C.c_int
        L.tmo_traits_init_from_samples.argtypes = [C.c_int, f32p, f32p, f32p]
        L.tmo_traits_project.restype = C.c_int
        L.tmo_traits_project.argtypes = [C.c_int, f32p, f32p, f32p]
        L.tmo_traits_unproject.argtypes = [C.c_int, f32p, f32p, f32p]
        L.tmo_traits_tangent.argtypes = [C.c_int, f32p, f32p, f32p]
        L.tmo_traits_normal.argtypes = [C.c_int, f32p, f32p, f32p]
        L.tmo_traits_intrinsic_distance.restype = C.c_float
        L.tmo_traits_intrinsic_distance.argtypes = [C.c_int, f32p, f32p, f32p]
        L.tmo_cl_uv_project.argtypes = [f32p, f32p, f32p]
        L.tmo_cl_icp_projection.argtypes = [f32p, C.c_int, f32p, i32p, i32p, f32p, f32p, f32p, f32p,
                                            C.c_float, f32p, i32p, i32p]
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def _u(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(u32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(i32p)


STATE_FLOATS = 37
IDENTITY, PLANE, PLANE2, CYLINDER = 0, 1, 2, 3


def cloud_resolution(surfels, threads=8):
    s, p = _f(np.asarray(surfels).reshape(-1, 12))
    d = np.zeros(s.shape[0], np.float32)
    r = lib().tmo_cloud_resolution(p, s.shape[0], d.ctypes.data_as(f32p), threads)
    return float(r), d


def cloud_curvature(surfels, k=30, threads=8):
    s, p = _f(np.asarray(surfels).reshape(-1, 12))
    out = np.zeros((s.shape[0], 5), np.float32)
    lib().tmo_cloud_curvature(p, s.shape[0], k, out.ctypes.data_as(f32p), threads)
    return out


def atan2_census(surfels, angle_step, subset=None, lower=0.0, upper=0.0, ij=None, cap=4096):
    """bins that change with glibc's atan2f in place of the project's: dict(pairs, bins_differ, bits_differ,
    max_abs_diff, differing_ij)"""
    s, p = _f(np.asarray(surfels).reshape(-1, 12))
    diff = np.zeros((cap, 2), np.uint32)
    n_pairs, n_bits = C.c_uint64(0), C.c_uint64(0)
    worst = np.zeros(1, np.float32)
    if ij is None:
        sub, sp = _u(subset)
        nb = lib().tmo_atan2_census(p, s.shape[0], sp, len(sub), None, 0, lower, upper, angle_step,
                                    diff.ctypes.data_as(u32p), cap, C.byref(n_pairs), C.byref(n_bits),
                                    worst.ctypes.data_as(f32p))
    else:
        pairs, pp = _u(np.asarray(ij).reshape(-1, 2))
        nb = lib().tmo_atan2_census(p, s.shape[0], None, 0, pp, pairs.shape[0], 0.0, 0.0, angle_step,
                                    diff.ctypes.data_as(u32p), cap, C.byref(n_pairs), C.byref(n_bits),
                                    worst.ctypes.data_as(f32p))
    return dict(pairs=int(n_pairs.value), bins_differ=int(nb), bits_differ=int(n_bits.value),
                max_abs_diff=float(worst[0]), differing_ij=diff[:min(nb, cap)].copy())


def early_drop_compare(subset_max, step_subset, step_tried, step_corrs):
    """(points compared, points where the restated bound differs from the literal transcription, first few)"""
    n = C.c_uint64(0)
    bad = np.zeros((16, 3), np.uint32)
    k = lib().tmo_early_drop_compare(subset_max, step_subset, step_tried, step_corrs, C.byref(n),
                                     bad.ctypes.data_as(u32p), 16)
    return int(n.value), int(k), bad[:min(k, 16)]


def cloud_knn(surfels, query, k, subset=None):
    """knn_inclusive(k, point): (indices, squared distances), order (d2, index)"""
    s, p = _f(np.asarray(surfels).reshape(-1, 12))
    q, qp = _f(np.asarray(query, np.float32).reshape(3))
    sub, sp, ns = (None, None, 0) if subset is None else (*_u(subset), len(subset))
    idx = np.zeros(k, np.int32)
    d2 = np.zeros(k, np.float32)
    n = lib().tmo_cloud_knn(p, s.shape[0], sp, ns, qp, k, idx.ctypes.data_as(i32p), d2.ctypes.data_as(f32p))
    return idx[:n], d2[:n]


def cloud_radius_search(surfels, query, radius, subset=None):
    s, p = _f(np.asarray(surfels).reshape(-1, 12))
    q, qp = _f(np.asarray(query, np.float32).reshape(3))
    sub, sp, ns = (None, None, 0) if subset is None else (*_u(subset), len(subset))
    n = lib().tmo_cloud_radius_search(p, s.shape[0], sp, ns, qp, radius, None, None, 0)
    idx = np.zeros(n, np.int32)
    d2 = np.zeros(n, np.float32)
    lib().tmo_cloud_radius_search(p, s.shape[0], sp, ns, qp, radius, idx.ctypes.data_as(i32p), d2.ctypes.data_as(f32p), n)
    return idx, d2


def cloud_curvature_of(surfels, p_idx, indices):
    s, p = _f(np.asarray(surfels).reshape(-1, 12))
    nb = np.ascontiguousarray(indices, np.int32)
    out = np.zeros(5, np.float32)
    lib().tmo_cloud_curvature_of(p, p_idx, nb.ctypes.data_as(i32p), len(nb), out.ctypes.data_as(f32p))
    return out


def murmur3_32(words, seed):
    w, p = _u(words)
    return int(lib().tmo_murmur3_32(p, len(w), seed))


def murmur_key(key):
    w, p = _u(key)
    return int(lib().tmo_murmur_key(p, len(w)))


def discretize_range(v, mn, rng, steps):
    return int(lib().tmo_discretize_range(v, mn, rng, steps))


def discretize_step(v, step):
    return int(lib().tmo_discretize_step(v, step))


def atan2f(y, x):
    return float(lib().tmo_atan2f(y, x))


def feature(s0, s1):
    a, pa = _f(s0)
    b, pb = _f(s1)
    out = np.zeros(4, np.float32)
    lib().tmo_feature(pa, pb, out.ctypes.data_as(f32p))
    return out


def discretize_feature(f, bmin, bmax, distance_step_count, angle_step):
    a, pa = _f(f)
    b, pb = _f(bmin)
    c, pc = _f(bmax)
    out = np.zeros(4, np.uint32)
    lib().tmo_discretize_feature(pa, pb, pc, distance_step_count, angle_step,
                                 out.ctypes.data_as(u32p))
    return out


def valid(f, bmin, bmax):
    a, pa = _f(f)
    b, pb = _f(bmin)
    c, pc = _f(bmax)
    return bool(lib().tmo_valid(pa, pb, pc))


def base_transform(src_i, src_j, src_t, tgt_i, tgt_j, tgt_t):
    arrs = [_f(x) for x in (src_i, src_j, src_t, tgt_i, tgt_j, tgt_t)]
    out = np.zeros(16, np.float32)
    lib().tmo_base_transform(*[p for _, p in arrs], out.ctypes.data_as(f32p))
    return out


def early_drop_upper(tried, subset, corrs):
    return int(lib().tmo_early_drop_upper(tried, subset, corrs))


def early_drop_upper_literal(tried, subset, corrs):
    return int(lib().tmo_early_drop_upper_literal(tried, subset, corrs))


def early_drop_checkpoints(subset_size):
    out = np.zeros(18, np.uint32)
    lib().tmo_early_drop_checkpoints(subset_size, out.ctypes.data_as(u32p))
    return out


class Model:
    def __init__(self, surfels, resolution, distance_step_count=20.0, angle_step=None):
        if angle_step is None:
            angle_step = float(np.float32(10.0) * np.float32(np.pi) / np.float32(180.0))
        self.surfels, p = _f(np.asarray(surfels).reshape(-1, 12))
        self.n = self.surfels.shape[0]
        self.h = lib().tmo_model_create(p, self.n, resolution, distance_step_count, angle_step)

    def __del__(self):
        if getattr(self, "h", None):
            lib().tmo_model_destroy(self.h)
            self.h = None

    def set_grid_subset(self, subset):
        sb, ps = _u(subset)
        rc = lib().tmo_model_set_grid_subset(self.h, ps, sb.size)
        if rc:
            raise RuntimeError("tmo_model_set_grid_subset rc=%d" % rc)

    def build_voxels(self, threads=8, brute=False):
        fn = lib().tmo_model_build_voxels_bruteforce if brute else lib().tmo_model_build_voxels
        rc = fn(self.h, threads)
        if rc:
            raise RuntimeError("tmo_model_build_voxels rc=%d" % rc)

    def set_voxels(self, voxels):
        v, p = _u(voxels)
        rc = lib().tmo_model_set_voxels(self.h, p, v.size)
        if rc:
            raise RuntimeError("tmo_model_set_voxels rc=%d" % rc)

    def build_pairs(self, subset, min_diameter_factor=0.2, max_diameter_factor=1.0):
        s, p = _u(subset)
        return int(lib().tmo_model_build_pairs(self.h, p, s.size, min_diameter_factor,
                                               max_diameter_factor))

    @property
    def diameter(self):
        return float(lib().tmo_model_diameter(self.h))

    def set_metric(self, kind, radius=0.0):
        lib().tmo_model_set_metric(self.h, kind, radius)

    def grid(self):
        ext = np.zeros(3, np.int32)
        tv = np.zeros(16, np.float32)
        lib().tmo_model_get_grid(self.h, ext.ctypes.data_as(i32p), tv.ctypes.data_as(f32p))
        return ext, tv

    def voxels(self):
        n = int(lib().tmo_model_voxel_count(self.h))
        ptr = lib().tmo_model_voxels(self.h)
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    def feature_bounds(self):
        a = np.zeros(4, np.float32)
        b = np.zeros(4, np.float32)
        lib().tmo_model_get_feature_bounds(self.h, a.ctypes.data_as(f32p), b.ctypes.data_as(f32p))
        return a, b

    def pair_count(self):
        return int(lib().tmo_model_pair_count(self.h))

    def valid_count(self):
        return int(lib().tmo_model_valid_count(self.h))

    def query(self, f, cap=1 << 20):
        a, pa = _f(f)
        total = C.c_uint32(0)
        rc = lib().tmo_model_query(self.h, pa, None, 0, C.byref(total))
        if rc:
            raise RuntimeError("Cannot query uninitialized model")
        n = min(total.value, cap)
        out = np.zeros((n, 2), np.uint32)
        lib().tmo_model_query(self.h, pa, out.ctypes.data_as(u32p), n, C.byref(total))
        return out, total.value

    def query_key(self, key, cap=1 << 20):
        k, pk = _u(key)
        total = C.c_uint32(0)
        rc = lib().tmo_model_query_key(self.h, pk, None, 0, C.byref(total))
        if rc:
            raise RuntimeError("Cannot query uninitialized model")
        n = min(total.value, cap)
        out = np.zeros((n, 2), np.uint32)
        lib().tmo_model_query_key(self.h, pk, out.ctypes.data_as(u32p), n, C.byref(total))
        return out, total.value

    def table_keys(self):
        n = int(lib().tmo_model_table_keys(self.h, None, None, 0))
        keys = np.zeros((n, 4), np.uint32)
        counts = np.zeros(n, np.uint32)
        lib().tmo_model_table_keys(self.h, keys.ctypes.data_as(u32p), counts.ctypes.data_as(u32p), n)
        return keys, counts

    def voxel_query(self, pos4):
        a, pa = _f(pos4)
        idx = C.c_uint32(0)
        ok = lib().tmo_model_voxel_query(self.h, pa, C.byref(idx))
        return idx.value if ok else None


class Scene:
    def __init__(self, surfels, tangent_mask):
        self.surfels, p = _f(np.asarray(surfels).reshape(-1, 12))
        self.n = self.surfels.shape[0]
        self.tangent_mask, pm = _i(tangent_mask)
        self.h = lib().tmo_scene_create(p, self.n, pm)

    def __del__(self):
        if getattr(self, "h", None):
            lib().tmo_scene_destroy(self.h)
            self.h = None

    def set_mask(self, mask):
        if mask is None:
            lib().tmo_scene_set_mask(self.h, None)
        else:
            m, p = _i(mask)
            lib().tmo_scene_set_mask(self.h, p)

    def project(self, model, pose16, accept_prob, dist_thres, subset=None, early_out=False):
        T, pT = _f(pose16)
        if subset is None:
            ps, ns = None, 0
            cap = self.n
        else:
            subset, ps = _i(subset)
            ns = subset.size
            cap = ns
        sc = np.zeros(cap, np.uint32)
        mc = np.zeros(cap, np.uint32)
        score = C.c_double(0)
        saved = C.c_uint32(0)
        n = lib().tmo_project(self.h, model.h, ps, ns, pT, accept_prob, dist_thres, int(early_out),
                              sc.ctypes.data_as(u32p), mc.ctypes.data_as(u32p), C.byref(score),
                              C.byref(saved))
        return sc[:n].copy(), mc[:n].copy(), score.value, saved.value

    def generate_hypotheses(self, model, ij_pairs, min_diameter_factor=0.2, max_diameter_factor=1.0,
                            force_up=False, query_limit=200):
        ij, pij = _u(np.asarray(ij_pairs).reshape(-1, 2))
        cnt = C.c_uint32(0)
        rc = lib().tmo_generate_hypotheses(self.h, model.h, pij, ij.shape[0], min_diameter_factor,
                                           max_diameter_factor, int(force_up), query_limit, None,
                                           None, 0, C.byref(cnt))
        if rc:
            raise RuntimeError("Cannot query uninitialized model")
        n = cnt.value
        poses = np.zeros((n, 16), np.float32)
        prov = np.zeros((n, 4), np.uint32)
        lib().tmo_generate_hypotheses(self.h, model.h, pij, ij.shape[0], min_diameter_factor,
                                      max_diameter_factor, int(force_up), query_limit,
                                      poses.ctypes.data_as(f32p), prov.ctypes.data_as(u32p), n,
                                      C.byref(cnt))
        return poses, prov

    def score_hypotheses(self, model, poses16, accept_prob, dist_thres, subset=None,
                         early_drop=False, threads=1):
        P, pP = _f(np.asarray(poses16).reshape(-1, 16))
        n = P.shape[0]
        if subset is None:
            ps, ns = None, 0
        else:
            subset, ps = _i(subset)
            ns = subset.size
        counts = np.zeros(n, np.uint32)
        scores = np.zeros(n, np.float64)
        ev = C.c_uint64(0)
        lib().tmo_score_hypotheses(self.h, model.h, pP, n, ps, ns, accept_prob, dist_thres,
                                   int(early_drop), threads, counts.ctypes.data_as(u32p),
                                   scores.ctypes.data_as(f64p), C.byref(ev))
        return counts, scores, ev.value

    def radius_subset(self, center_idx, radius):
        n = lib().tmo_scene_radius_subset(self.h, center_idx, radius, None, 0)
        out = np.zeros(n, np.int32)
        lib().tmo_scene_radius_subset(self.h, center_idx, radius, out.ctypes.data_as(i32p), n)
        return out


def traits_init_from_samples(kind, model_state, samples):
    ms, pm = _f(model_state)
    s, ps = _f(samples)
    st = np.zeros(STATE_FLOATS, np.float32)
    rej = lib().tmo_traits_init_from_samples(kind, pm, ps, st.ctypes.data_as(f32p))
    return None if rej else st


def traits_project(kind, state, xyz):
    s, ps = _f(state)
    x, px = _f(xyz)
    out = np.zeros(3, np.float32)
    ok = lib().tmo_traits_project(kind, ps, px, out.ctypes.data_as(f32p))
    return out if ok else None


def traits_unproject(kind, state, uvw):
    s, ps = _f(state)
    x, px = _f(uvw)
    out = np.zeros(3, np.float32)
    lib().tmo_traits_unproject(kind, ps, px, out.ctypes.data_as(f32p))
    return out


def traits_tangent(kind, state, surfel):
    s, ps = _f(state)
    x, px = _f(surfel)
    out = np.zeros(3, np.float32)
    lib().tmo_traits_tangent(kind, ps, px, out.ctypes.data_as(f32p))
    return out


def traits_normal(kind, state, surfel):
    s, ps = _f(state)
    x, px = _f(surfel)
    out = np.zeros(3, np.float32)
    lib().tmo_traits_normal(kind, ps, px, out.ctypes.data_as(f32p))
    return out


def traits_intrinsic_distance(kind, state, p0, p1):
    s, ps = _f(state)
    a, pa = _f(p0)
    b, pb = _f(p1)
    return float(lib().tmo_traits_intrinsic_distance(kind, ps, pa, pb))
