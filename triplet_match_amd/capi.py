"""ctypes binding of libtm_hip.so (include/tm_hip.h) with numpy-facing wrappers.

This is plumbing for tests, bench.py and smoke(): the product is the C ABI itself. There is no
fallback of any kind — if the HIP library is missing or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TM_LIB selects another build of the same ABI (libtm_hip_experiments.so of `build.py --timing-experiments`)
LIB_PATH = os.environ.get("TM_LIB") or os.path.join(_HERE, "libtm_hip.so")

OK, ERR_INVALID, ERR_NOT_INITIALIZED, ERR_HIP, ERR_EMPTY, ERR_UNSUPPORTED, ERR_NO_DEVICE, \
    ERR_CAPACITY = 0, -1, -2, -3, -4, -5, -6, -7
IDENTITY, PLANE, PLANE2, CYLINDER = 0, 1, 2, 3
STATE_FLOATS = 37
SCORE_AUTO, SCORE_BRUTEFORCE, SCORE_CULLED = 0, 1, 2

u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


class BestRecord(C.Structure):
    _fields_ = [("inlier_count", C.c_uint32), ("hyp_id", C.c_uint32), ("score", C.c_double),
                ("pose", C.c_float * 16), ("pad", C.c_uint32 * 4)]


class BatchResult(C.Structure):
    """tm_batch_result (include/tm_hip.h), 88 bytes"""
    _fields_ = [("index", C.c_int32), ("n_hypotheses", C.c_uint32), ("count", C.c_uint32),
                ("dropped_at", C.c_int32), ("score", C.c_double), ("pose", C.c_float * 16)]


assert C.sizeof(BestRecord) == 96

# every symbol include/tm_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "tm_last_error": (C.c_char_p, []),
    "tm_version": (C.c_char_p, []),
    "tm_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "tm_ctx_destroy": (C.c_int, [vp]),
    "tm_ctx_synchronize": (C.c_int, [vp]),
    "tm_ctx_pool_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "tm_ctx_stream": (vp, [vp]),
    "tm_device_alloc": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "tm_device_free": (C.c_int, [vp, vp]),
    "tm_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "tm_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "tm_ctx_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "tm_ctx_get_option": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int)]),
    "tm_ctx_time_begin": (C.c_int, [vp]),
    "tm_ctx_time_end": (C.c_int, [vp, f32p]),
    "tm_ctx_enable_timing": (C.c_int, [vp, C.c_int]),
    "tm_ctx_last_score_timing": (C.c_int, [vp, f32p, u64p, u64p, u64p]),
    "tm_ctx_score_timing_totals": (C.c_int, [vp, u32p, f32p, u64p]),
    "tm_murmur3_32": (C.c_uint32, [u32p, C.c_int, C.c_uint32]),
    "tm_hash_key": (C.c_uint32, [u32p, C.c_int]),
    "tm_discretize_range": (C.c_uint32, [C.c_float, C.c_float, C.c_float, C.c_uint32]),
    "tm_discretize_step": (C.c_uint32, [C.c_float, C.c_float]),
    "tm_atan2f": (C.c_float, [C.c_float, C.c_float]),
    "tm_feature": (None, [f32p, f32p, f32p]),
    "tm_discretize_feature": (None, [f32p, f32p, f32p, C.c_float, C.c_float, u32p]),
    "tm_valid": (C.c_int, [f32p, f32p, f32p]),
    "tm_valid_bounds": (None, [f32p, f32p, C.c_float, C.c_float, f32p, f32p]),
    "tm_base_transform": (None, [f32p] * 7),
    "tm_early_drop_upper": (C.c_uint32, [C.c_uint32] * 3),
    "tm_features_batch": (C.c_int, [vp, f32p, C.c_uint32, u32p, C.c_uint32, f32p, f32p, C.c_float,
                                    C.c_float, f32p, u32p, u32p]),
    "tm_cloud_resolution": (C.c_int, [vp, f32p, C.c_uint32, f32p, f32p]),
    "tm_cloud_curvature": (C.c_int, [vp, f32p, C.c_uint32, C.c_uint32, f32p]),
    "tm_cloud_curvature_subset": (C.c_int, [vp, f32p, C.c_uint32, C.c_uint32, u32p, C.c_uint32, f32p]),
    "tm_scene_curvature_subset": (C.c_int, [vp, C.c_uint32, u32p, C.c_uint32, f32p]),
    "tm_scene_set_tangent_mask": (C.c_int, [vp, i32p]),
    "tm_scene_auto_tangent_mask": (C.c_int, [vp, C.c_uint32, C.c_float, C.c_float, C.c_void_p, i32p, u32p, f32p, C.c_uint32,
                                             u32p, u32p]),
    "tm_cloud_create": (C.c_int, [vp, f32p, C.c_uint32, u32p, C.c_uint32, C.POINTER(vp)]),
    "tm_cloud_destroy": (C.c_int, [vp]),
    "tm_cloud_tree_size": (C.c_uint32, [vp]),
    "tm_cloud_knn": (C.c_int, [vp, f32p, C.c_uint32, C.c_uint32, i32p, f32p, u32p]),
    "tm_cloud_radius_search": (C.c_int, [vp, f32p, C.c_float, i32p, f32p, C.c_uint32, u32p]),
    "tm_cloud_curvature_of": (C.c_int, [vp, C.c_uint32, i32p, C.c_uint32, f32p]),
    "tm_model_create": (C.c_int, [vp, f32p, C.c_uint32, C.c_float, C.c_float, C.c_float,
                                  C.POINTER(vp)]),
    "tm_model_destroy": (C.c_int, [vp]),
    "tm_model_set_grid_subset": (C.c_int, [vp, u32p, C.c_uint32]),
    "tm_model_build_voxels": (C.c_int, [vp]),
    "tm_model_build_pairs": (C.c_int, [vp, u32p, C.c_uint32, C.c_float, C.c_float]),
    "tm_model_set_metric": (C.c_int, [vp, C.c_int, C.c_float]),
    "tm_model_get_info": (C.c_int, [vp, f32p, u32p, u32p, f32p]),
    "tm_model_get_grid": (C.c_int, [vp, i32p, f32p]),
    "tm_model_download_voxels": (C.c_int, [vp, u32p, C.c_uint64, u64p]),
    "tm_model_get_feature_bounds": (C.c_int, [vp, f32p, f32p]),
    "tm_model_get_pair_counts": (C.c_int, [vp, u64p, u64p, u32p]),
    "tm_model_download_table": (C.c_int, [vp, u32p, u32p, C.c_uint32, u32p, C.c_uint64]),
    "tm_model_query": (C.c_int, [vp, f32p, u32p, C.c_uint32, u32p]),
    "tm_model_query_key": (C.c_int, [vp, u32p, u32p, C.c_uint32, u32p]),
    "tm_model_voxel_query": (C.c_int, [vp, f32p, u32p, C.POINTER(C.c_int)]),
    "tm_scene_create": (C.c_int, [vp, f32p, C.c_uint32, i32p, C.POINTER(vp)]),
    "tm_scene_create_dev": (C.c_int, [vp, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(vp)]),
    "tm_scene_destroy": (C.c_int, [vp]),
    "tm_scene_set_mask": (C.c_int, [vp, i32p]),
    "tm_scene_mask_add": (C.c_int, [vp, u32p, C.c_uint32]),
    "tm_scene_mask_add_extracted": (C.c_int, [vp, u32p, C.c_uint32]),
    "tm_extract_arbitrate_batch": (C.c_int, [vp, vp, f32p, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, u32p, u32p,
                                             C.c_uint32, u32p, f64p, i32p, u32p, u32p]),
    "tm_scene_size": (C.c_uint32, [vp]),
    "tm_scene_create_projected": (C.c_int, [vp, C.c_int, f32p, C.POINTER(vp)]),
    "tm_scene_probe_projected": (C.c_int, [vp, C.c_int, f32p, C.c_float, u64p, u32p]),
    "tm_scene_download": (C.c_int, [vp, f32p, i32p]),
    "tm_scene_tangent_points": (C.c_int, [vp, u32p, f32p, C.c_uint32, u32p]),
    "tm_scene_radius_subset": (C.c_int, [vp, C.c_uint32, C.c_float, i32p, C.c_uint32, u32p]),
    "tm_scene_radius_select": (C.c_int, [vp, u32p, C.c_uint32, i32p]),
    "tm_scene_radius_lists": (C.c_int, [vp, u32p, C.c_uint32, C.c_float, i32p, C.c_uint32, u32p]),
    "tm_scene_radius_tangent_lists": (C.c_int, [vp, u32p, C.c_uint32, C.c_float, u32p, u32p, u32p, C.c_uint32, u32p]),
    "tm_pair_is_valid_sample": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_float, C.c_float]),
    "tm_pair_valid_raw": (C.c_int, [f32p, f32p, C.c_float, C.c_float, C.c_float, f32p, f32p]),
    "tm_generate_hypotheses": (C.c_int, [vp, vp, u32p, C.c_uint32, C.c_float, C.c_float, C.c_int,
                                         C.c_int, f32p, u32p, C.c_uint32, u32p, u32p]),
    "tm_generate_hypotheses_dev": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_float, C.c_float, C.c_int,
                                             C.c_int, vp, vp, C.c_uint32, u32p, u32p]),
    "tm_score_hypotheses": (C.c_int, [vp, vp, C.c_int, f32p, C.c_uint32, i32p, C.c_uint32,
                                      C.c_float, C.c_int, u32p, f64p]),
    "tm_score_hypotheses_dev": (C.c_int, [vp, vp, C.c_int, vp, C.c_uint32, C.c_float, C.c_int, vp,
                                          vp]),
    "tm_score_hypotheses_ball": (C.c_int, [vp, vp, f32p, C.c_uint32, u32p, C.c_float, C.c_float, u32p, f64p]),
    "tm_score_hypotheses_ball_dev": (C.c_int, [vp, vp, vp, C.c_uint32, vp, C.c_float, C.c_float, vp, vp]),
    "tm_score_hypotheses_on_points": (C.c_int, [vp, vp, f32p, C.c_uint32, i32p, C.c_uint32, C.c_float, u32p, f64p]),
    "tm_score_hypotheses_early_drop": (C.c_int, [vp, vp, f32p, C.c_uint32, i32p, C.c_uint32,
                                                 C.c_float, C.c_float, u32p, f64p, i32p, u32p]),
    "tm_score_hypotheses_early_drop_ball": (C.c_int, [vp, vp, f32p, C.c_uint32, u32p, C.c_float, C.c_float, C.c_float,
                                                     u32p, f64p, i32p, u32p]),
    "tm_score_hypotheses_early_drop_ball_dev": (C.c_int, [vp, vp, vp, C.c_uint32, u32p, C.c_float, C.c_float, C.c_float,
                                                         vp, vp, vp, vp]),
    "tm_search_batch": (C.c_int, [vp, vp, u32p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float,
                                  C.c_float, C.c_float, C.c_int, C.c_float, C.POINTER(BatchResult)]),
    "tm_search_batches": (C.c_int, [vp, vp, u32p, C.c_uint32, u32p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int,
                                    C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.POINTER(BatchResult)]),
    "tm_extract_correspondences": (C.c_int, [vp, vp, f32p, C.c_float, u32p, u32p, C.c_uint32, u32p,
                                             f64p]),
    "tm_extract_correspondences_batch": (C.c_int, [vp, vp, f32p, C.c_uint32, C.c_float, u32p, u32p, C.c_uint32, u32p,
                                                   f64p]),
    "tm_icp_accumulate": (C.c_int, [vp, vp, f32p, C.c_float, u32p, C.POINTER(C.c_longlong)]),
    "tm_icp_refine": (C.c_int, [vp, vp, f32p, C.c_float, C.c_uint32, f32p, u32p, u32p]),
    "tm_icp_refine_batch": (C.c_int, [vp, vp, f32p, C.c_uint32, C.c_float, C.c_uint32, f32p, u32p, u32p]),
    "tm_rigid_fit": (C.c_int, [C.POINTER(C.c_longlong), C.c_uint32, f32p]),
    "tm_best_of_batch": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.POINTER(BestRecord)]),
    "tm_scene_live_count": (C.c_uint64, [vp]),
    "tm_best_select": (C.c_int, [C.POINTER(BestRecord), C.c_uint32, C.POINTER(BestRecord)]),
    "tm_comm_get_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "tm_comm_create": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.POINTER(vp)]),
    "tm_comm_destroy": (C.c_int, [vp]),
    "tm_comm_rank": (C.c_int, [vp]),
    "tm_comm_world": (C.c_int, [vp]),
    "tm_best_of_batch_prov": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]),
    "tm_best_reduce": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(BestRecord)]),
    "tm_best_reduce_begin": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(C.c_int)]),
    "tm_best_reduce_end": (C.c_int, [vp, C.c_int, C.POINTER(BestRecord)]),
    "tm_comm_probe": (C.c_int, []),
    "tm_comm_allgather_records": (C.c_int, [vp, C.POINTER(BestRecord), C.c_uint32, C.POINTER(BestRecord)]),
    "tm_traits_init_from_samples": (C.c_int, [C.c_int, f32p, f32p, f32p]),
    "tm_traits_project": (C.c_int, [C.c_int, f32p, f32p, f32p]),
    "tm_traits_unproject": (None, [C.c_int, f32p, f32p, f32p]),
    "tm_traits_tangent": (None, [C.c_int, f32p, f32p, f32p]),
    "tm_traits_normal": (None, [C.c_int, f32p, f32p, f32p]),
    "tm_traits_intrinsic_distance": (C.c_float, [C.c_int, f32p, f32p, f32p]),
    "tm_traits_project_cloud": (C.c_int, [vp, C.c_int, f32p, f32p, C.c_uint32, f32p, i32p]),
    "tm_cl_icp_projection": (C.c_int, [vp, f32p, C.c_int, f32p, i32p, i32p, f32p, f32p, f32p, f32p, C.c_float,
                                       f32p, i32p, i32p, u32p]),
    "tm_cl_icp_correlation": (C.c_int, [vp, f32p, C.c_uint32, f32p, C.c_uint32, i32p, i32p, C.c_int, f32p, f32p,
                                        f64p]),
}

_lib = None


class TmError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("tm error %d: %s" % (code, text))
        self.code = code


def lib():
    """Load libtm_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libtm_hip.so is not built: run `python -m triplet_match_amd.build` "
                               "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != OK:
        raise TmError(rc, lib().tm_last_error().decode(errors="replace"))


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def _u(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(u32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(i32p)


def default_angle_step():
    """10 degrees in radians computed in f32 (reference apps/triplet_match.cpp:19)."""
    return float(np.float32(10.0) * np.float32(np.pi) / np.float32(180.0))


# ---- host scalar functions -------------------------------------------------------------------
def murmur3_32(words, seed):
    w, p = _u(words)
    return int(lib().tm_murmur3_32(p, len(w), seed))


def hash_key(key):
    w, p = _u(key)
    return int(lib().tm_hash_key(p, len(w)))


def discretize_range(v, mn, rng, steps):
    return int(lib().tm_discretize_range(v, mn, rng, steps))


def discretize_step(v, step):
    return int(lib().tm_discretize_step(v, step))


def atan2f(y, x):
    return float(lib().tm_atan2f(y, x))


def feature(s0, s1):
    a, pa = _f(s0)
    b, pb = _f(s1)
    out = np.zeros(4, np.float32)
    lib().tm_feature(pa, pb, out.ctypes.data_as(f32p))
    return out


def discretize_feature(f, bmin, bmax, distance_step_count, angle_step):
    a, pa = _f(f)
    b, pb = _f(bmin)
    c, pc = _f(bmax)
    out = np.zeros(4, np.uint32)
    lib().tm_discretize_feature(pa, pb, pc, distance_step_count, angle_step,
                                out.ctypes.data_as(u32p))
    return out


def valid(f, bmin, bmax):
    a, pa = _f(f)
    b, pb = _f(bmin)
    c, pc = _f(bmax)
    return bool(lib().tm_valid(pa, pb, pc))


def base_transform(src_i, src_j, src_t, tgt_i, tgt_j, tgt_t):
    arrs = [_f(x) for x in (src_i, src_j, src_t, tgt_i, tgt_j, tgt_t)]
    out = np.zeros(16, np.float32)
    lib().tm_base_transform(*[p for _, p in arrs], out.ctypes.data_as(f32p))
    return out


def early_drop_upper(tried, subset, corrs):
    return int(lib().tm_early_drop_upper(tried, subset, corrs))


# ---- handles -----------------------------------------------------------------------------------
class Context:
    def __init__(self, device=0):
        h = vp()
        _check(lib().tm_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            lib().tm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except TypeError:  # interpreter shutdown: the module's globals are gone already, and the process is ending
            pass

    def synchronize(self):
        _check(lib().tm_ctx_synchronize(self.h))

    def pool_stats(self):
        """tm_ctx_pool_stats: (cached bytes, requests served from the cache, requests served by hipMalloc)"""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().tm_ctx_pool_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_option(self, name, value):
        """tm_ctx_set_option: a cross-check / timing switch of this context (never the environment)."""
        _check(lib().tm_ctx_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int(0)
        _check(lib().tm_ctx_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def option(self, name, value):
        """with ctx.option("score_grouped", 0): ... — sets the switch for the block and restores it."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            old = self.get_option(name)
            self.set_option(name, value)
            try:
                yield
            finally:
                self.set_option(name, old)
        return _cm()

    def time_begin(self):
        _check(lib().tm_ctx_time_begin(self.h))

    def time_end(self):
        """milliseconds of GPU time since time_begin (HIP events on the context's stream)"""
        ms = C.c_float(0)
        _check(lib().tm_ctx_time_end(self.h, C.byref(ms)))
        return ms.value

    def enable_timing(self, on=True):
        _check(lib().tm_ctx_enable_timing(self.h, int(on)))

    def last_score_timing(self):
        ms = C.c_float(0)
        ev, ps, hy = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().tm_ctx_last_score_timing(self.h, C.byref(ms), C.byref(ev), C.byref(ps),
                                              C.byref(hy)))
        return dict(ms=ms.value, evaluations=ev.value, passing=ps.value, hypotheses=hy.value)

    def score_timing_totals(self):
        """Sums over the scoring launches since enable_timing(True): kernel ms (HIP events on the library's
        stream) and the kernels' own counters. Synchronises the stream."""
        n, ms = C.c_uint32(0), C.c_float(0)
        tot = (C.c_uint64 * 6)()
        _check(lib().tm_ctx_score_timing_totals(self.h, C.byref(n), C.byref(ms), tot))
        return dict(launches=n.value, ms=ms.value, evaluations=tot[0], passing=tot[1], in_grid=tot[2],
                    near=tot[3], hypotheses=tot[4], staged_chunks=tot[5])

    def alloc(self, nbytes):
        p = vp()
        _check(lib().tm_device_alloc(self.h, nbytes, C.byref(p)))
        return p

    def free(self, p):
        _check(lib().tm_device_free(self.h, p))

    def h2d(self, d_ptr, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib().tm_memcpy_h2d(self.h, d_ptr, arr.ctypes.data_as(vp), arr.nbytes))

    def d2h(self, arr, d_ptr):
        assert arr.flags["C_CONTIGUOUS"]
        _check(lib().tm_memcpy_d2h(self.h, arr.ctypes.data_as(vp), d_ptr, arr.nbytes))

    def features_batch(self, surfels, ab_pairs, bmin=None, bmax=None, distance_step_count=20.0,
                       angle_step=None):
        if angle_step is None:
            angle_step = default_angle_step()
        s, ps = _f(np.asarray(surfels).reshape(-1, 12))
        ab, pab = _u(np.asarray(ab_pairs).reshape(-1, 2))
        n = ab.shape[0]
        feats = np.zeros((n, 4), np.float32)
        keys = np.zeros((n, 4), np.uint32)
        hashes = np.zeros(n, np.uint32)
        if bmin is not None:
            b0, pb0 = _f(bmin)
            b1, pb1 = _f(bmax)
        else:
            pb0 = pb1 = None
        _check(lib().tm_features_batch(self.h, ps, s.shape[0], pab, n, pb0, pb1,
                                       distance_step_count, angle_step,
                                       feats.ctypes.data_as(f32p), keys.ctypes.data_as(u32p),
                                       hashes.ctypes.data_as(u32p)))
        return feats, keys, hashes

    def cloud_resolution(self, surfels):
        s, ps = _f(np.asarray(surfels).reshape(-1, 12))
        res = C.c_float(0)
        d = np.zeros(s.shape[0], np.float32)
        _check(lib().tm_cloud_resolution(self.h, ps, s.shape[0], C.byref(res), d.ctypes.data_as(f32p)))
        return res.value, d

    def cloud_curvature(self, surfels, k=30):
        s, ps = _f(np.asarray(surfels).reshape(-1, 12))
        out = np.zeros((s.shape[0], 5), np.float32)
        _check(lib().tm_cloud_curvature(self.h, ps, s.shape[0], k, out.ctypes.data_as(f32p)))
        return out

    def cloud_curvature_subset(self, surfels, indices, k=30):
        s, ps = _f(np.asarray(surfels).reshape(-1, 12))
        idx, pi = _u(indices)
        out = np.zeros((len(idx), 5), np.float32)
        _check(lib().tm_cloud_curvature_subset(self.h, ps, s.shape[0], k, pi, len(idx), out.ctypes.data_as(f32p)))
        return out

    def best_of_batch(self, d_counts, d_scores, d_poses, n_hyp, id_offset=0, id_stride=1):
        rec = BestRecord()
        _check(lib().tm_best_of_batch(self.h, d_counts, d_scores, d_poses, n_hyp, id_offset,
                                      id_stride, C.byref(rec)))
        return rec

    def best_of_batch_prov(self, d_counts, d_scores, d_poses, d_prov, n_hyp, d_best, pair_offset=0, pair_stride=1,
                           query_limit=200):
        """Local argmax with canonical ids (global pair index * query_limit + entry); the record is written
        to the device buffer d_best (96 bytes) and stays there for Comm.best_reduce."""
        _check(lib().tm_best_of_batch_prov(self.h, d_counts, d_scores, d_poses, d_prov, n_hyp, pair_offset,
                                           pair_stride, query_limit, d_best))

    def traits_project_cloud(self, kind, state, surfels):
        st, pst = _f(state)
        s, ps = _f(np.asarray(surfels).reshape(-1, 12))
        out = np.zeros_like(s)
        valid_ = np.zeros(s.shape[0], np.int32)
        _check(lib().tm_traits_project_cloud(self.h, kind, pst, ps, s.shape[0],
                                             out.ctypes.data_as(f32p), valid_.ctypes.data_as(i32p)))
        return out, valid_


def cl_icp_projection(ctx, pnts4, image4, img_size, img_margin, mat_align, mat_uvw, mat_proj, mat_norm,
                      max_corr_dist):
    p, pp = _f(np.asarray(pnts4).reshape(-1, 4))
    im, pim = _f(np.asarray(image4).reshape(-1, 4))
    sz, psz = _i(img_size)
    mg, pmg = _i(img_margin)
    mats = [_f(m) for m in (mat_align, mat_uvw, mat_proj, mat_norm)]
    n = p.shape[0]
    out = np.zeros((n, 4), np.float32)
    mi = np.zeros(n, np.int32)
    si = np.zeros(n, np.int32)
    cnt = C.c_uint32(0)
    _check(lib().tm_cl_icp_projection(ctx.h, pp, n, pim, psz, pmg, mats[0][1], mats[1][1], mats[2][1], mats[3][1],
                                      max_corr_dist, out.ctypes.data_as(f32p), mi.ctypes.data_as(i32p),
                                      si.ctypes.data_as(i32p), C.byref(cnt)))
    return out, mi, si, cnt.value


def cl_icp_correlation(ctx, scene4, model4, idx_scene, idx_model, centroid_scene, centroid_model):
    s, ps = _f(np.asarray(scene4).reshape(-1, 4))
    m, pm = _f(np.asarray(model4).reshape(-1, 4))
    a, pa = _i(idx_scene)
    b, pb = _i(idx_model)
    cs, pcs = _f(centroid_scene)
    cm, pcm = _f(centroid_model)
    cov = np.zeros(9, np.float64)
    _check(lib().tm_cl_icp_correlation(ctx.h, ps, s.shape[0], pm, m.shape[0], pa, pb, a.size, pcs, pcm,
                                       cov.ctypes.data_as(f64p)))
    return cov


COMM_ID_BYTES = 128


def comm_unique_id():
    """128 bytes rank 0 creates and hands to the other ranks (any channel) before Comm(...)."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    _check(lib().tm_comm_get_unique_id(buf))
    return bytes(buf)


def comm_probe():
    """True when RCCL can be used by this process (tm_comm_probe); ranks exchange this before Comm(...)."""
    return lib().tm_comm_probe() == OK


class PendingReduce:
    """A tm_best_reduce in flight (tm_best_reduce_begin); result() = tm_best_reduce_end."""

    def __init__(self, ctx, ticket, n_models):
        self.ctx, self.ticket, self.n = ctx, ticket, n_models
        self._out = None

    def result(self):
        if self._out is None:
            out = (BestRecord * self.n)()
            _check(lib().tm_best_reduce_end(self.ctx.h, self.ticket, out))
            self._out = [BestRecord.from_buffer_copy(bytes(out[k])) for k in range(self.n)]
        return self._out


def best_reduce_begin(ctx, comm, d_local, n_models=1):
    """Queues all-gather (comm may be None = single process) + selection + copy to pinned memory; no wait."""
    t = C.c_int(-1)
    _check(lib().tm_best_reduce_begin(ctx.h, comm.h if comm is not None else None, d_local, n_models, C.byref(t)))
    return PendingReduce(ctx, t.value, n_models)


class Comm:
    """RCCL communicator of the C ABI (tm_comm): one per process / GPU. Collective constructor."""

    def __init__(self, ctx, rank=0, world=1, unique_id=None):
        self.ctx = ctx
        self.h = vp()
        idp = None
        if unique_id is not None:
            assert len(unique_id) == COMM_ID_BYTES
            idp = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib().tm_comm_create(ctx.h, rank, world, idp, C.byref(self.h)))
        self.rank, self.world = rank, world

    def close(self):
        if getattr(self, "h", None):
            lib().tm_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def best_reduce(self, d_local, n_models=1):
        """d_local: device pointer to this rank's n_models records. Returns the global winners (host)."""
        out = (BestRecord * n_models)()
        _check(lib().tm_best_reduce(self.ctx.h, self.h, d_local, n_models, out))
        return [BestRecord.from_buffer_copy(bytes(out[k])) for k in range(n_models)]

    def best_reduce_begin(self, d_local, n_models=1):
        return best_reduce_begin(self.ctx, self, d_local, n_models)

    def allgather_records(self, records):
        n = len(records)
        loc = (BestRecord * n)(*records)
        out = (BestRecord * (n * self.world))()
        _check(lib().tm_comm_allgather_records(self.h, loc, n, out))
        return [BestRecord.from_buffer_copy(bytes(out[k])) for k in range(n * self.world)]


def best_select(records):
    arr = (BestRecord * len(records))(*records)
    out = BestRecord()
    _check(lib().tm_best_select(arr, len(records), C.byref(out)))
    return out


class Cloud:
    """tm_cloud: the neighbour queries of pointcloud<Point> (kd-tree on the device)."""

    def __init__(self, ctx, surfels, subset=None):
        self.ctx = ctx
        s, p = _f(surfels)
        self.n = s.shape[0]
        self.h = vp()
        if subset is None:
            _check(lib().tm_cloud_create(ctx.h, p, self.n, None, 0, C.byref(self.h)))
        else:
            sub, sp = _u(subset)
            _check(lib().tm_cloud_create(ctx.h, p, self.n, sp, len(sub), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            lib().tm_cloud_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def tree_size(self):
        return int(lib().tm_cloud_tree_size(self.h))

    def knn(self, queries_xyz, k):
        """(indices [nq, k] int32, squared distances [nq, k] f32, counts [nq])"""
        q, qp = _f(np.asarray(queries_xyz, np.float32).reshape(-1, 3))
        nq = q.shape[0]
        idx = np.zeros((nq, k), np.int32)
        d2 = np.zeros((nq, k), np.float32)
        cnt = np.zeros(nq, np.uint32)
        _check(lib().tm_cloud_knn(self.h, qp, nq, k, idx.ctypes.data_as(i32p), d2.ctypes.data_as(f32p),
                                  cnt.ctypes.data_as(u32p)))
        return idx, d2, cnt

    def radius_search(self, query_xyz, radius):
        q, qp = _f(np.asarray(query_xyz, np.float32).reshape(3))
        n = C.c_uint32(0)
        _check(lib().tm_cloud_radius_search(self.h, qp, radius, None, None, 0, C.byref(n)))
        idx = np.zeros(n.value, np.int32)
        d2 = np.zeros(n.value, np.float32)
        if n.value:
            _check(lib().tm_cloud_radius_search(self.h, qp, radius, idx.ctypes.data_as(i32p), d2.ctypes.data_as(f32p),
                                                n.value, C.byref(n)))
        return idx, d2

    def curvature_of(self, idx, neighbours):
        nb, nbp = _i(neighbours)
        out = np.zeros(5, np.float32)
        _check(lib().tm_cloud_curvature_of(self.h, idx, nbp, len(nb), out.ctypes.data_as(f32p)))
        return out


class Model:
    """model<Point> over the C ABI (reference include/model:15-66)."""

    def __init__(self, ctx, surfels, resolution, distance_step_count=20.0, angle_step=None):
        if angle_step is None:
            angle_step = default_angle_step()
        self.ctx = ctx
        s, p = _f(np.asarray(surfels).reshape(-1, 12))
        self.n = s.shape[0]
        h = vp()
        _check(lib().tm_model_create(ctx.h, p, self.n, resolution, distance_step_count, angle_step,
                                     C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().tm_model_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_grid_subset(self, subset=None):
        """bounding box / diameter / grid over these points (include/impl/model.hpp:17-38); None = whole cloud"""
        if subset is None or len(subset) == 0:
            _check(lib().tm_model_set_grid_subset(self.h, None, 0))
        else:
            sb, ps = _u(subset)
            _check(lib().tm_model_set_grid_subset(self.h, ps, sb.size))

    def build_voxels(self):
        _check(lib().tm_model_build_voxels(self.h))

    def build_pairs(self, subset, min_diameter_factor=0.2, max_diameter_factor=1.0):
        s, p = _u(subset)
        _check(lib().tm_model_build_pairs(self.h, p, s.size, min_diameter_factor,
                                          max_diameter_factor))

    def set_metric(self, kind, radius=0.0):
        _check(lib().tm_model_set_metric(self.h, kind, radius))

    def info(self):
        d, r = C.c_float(0), C.c_float(0)
        n, pc = C.c_uint32(0), C.c_uint32(0)
        _check(lib().tm_model_get_info(self.h, C.byref(d), C.byref(n), C.byref(pc), C.byref(r)))
        return dict(diameter=d.value, cloud_size=n.value, point_count=pc.value, resolution=r.value)

    @property
    def diameter(self):
        return self.info()["diameter"]

    def grid(self):
        ext = np.zeros(3, np.int32)
        tv = np.zeros(16, np.float32)
        _check(lib().tm_model_get_grid(self.h, ext.ctypes.data_as(i32p), tv.ctypes.data_as(f32p)))
        return ext, tv

    def voxels(self):
        cnt = C.c_uint64(0)
        _check(lib().tm_model_download_voxels(self.h, None, 0, C.byref(cnt)))
        out = np.zeros(cnt.value, np.uint32)
        _check(lib().tm_model_download_voxels(self.h, out.ctypes.data_as(u32p), out.size,
                                              C.byref(cnt)))
        return out

    def feature_bounds(self):
        a = np.zeros(4, np.float32)
        b = np.zeros(4, np.float32)
        _check(lib().tm_model_get_feature_bounds(self.h, a.ctypes.data_as(f32p),
                                                 b.ctypes.data_as(f32p)))
        return a, b

    def pair_counts(self):
        e, v, k = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        _check(lib().tm_model_get_pair_counts(self.h, C.byref(e), C.byref(v), C.byref(k)))
        return dict(entries=e.value, valid_pairs=v.value, distinct_keys=k.value)

    def table(self):
        pc = self.pair_counts()
        keys = np.zeros((pc["distinct_keys"], 4), np.uint32)
        offs = np.zeros(pc["distinct_keys"] + 1, np.uint32)
        pairs = np.zeros((pc["entries"], 2), np.uint32)
        _check(lib().tm_model_download_table(self.h, keys.ctypes.data_as(u32p),
                                             offs.ctypes.data_as(u32p), keys.shape[0],
                                             pairs.ctypes.data_as(u32p), pairs.shape[0]))
        return keys, offs, pairs

    def query(self, f, cap=1 << 20):
        a, pa = _f(f)
        total = C.c_uint32(0)
        _check(lib().tm_model_query(self.h, pa, None, 0, C.byref(total)))
        n = min(total.value, cap)
        out = np.zeros((n, 2), np.uint32)
        _check(lib().tm_model_query(self.h, pa, out.ctypes.data_as(u32p), n, C.byref(total)))
        return out, total.value

    def query_key(self, key, cap=1 << 20):
        k, pk = _u(key)
        total = C.c_uint32(0)
        _check(lib().tm_model_query_key(self.h, pk, None, 0, C.byref(total)))
        n = min(total.value, cap)
        out = np.zeros((n, 2), np.uint32)
        _check(lib().tm_model_query_key(self.h, pk, out.ctypes.data_as(u32p), n, C.byref(total)))
        return out, total.value

    def voxel_query(self, pos4):
        a, pa = _f(pos4)
        idx = C.c_uint32(0)
        found = C.c_int(0)
        _check(lib().tm_model_voxel_query(self.h, pa, C.byref(idx), C.byref(found)))
        return idx.value if found.value else None


class Scene:
    """scene<Point> over the C ABI (reference include/scene:10-59)."""

    def __init__(self, ctx, surfels, tangent_mask):
        self.ctx = ctx
        s, p = _f(np.asarray(surfels).reshape(-1, 12))
        self.n = s.shape[0]
        tm, ptm = _i(tangent_mask)
        h = vp()
        _check(lib().tm_scene_create(ctx.h, p, self.n, ptm, C.byref(h)))
        self.h = h

    @classmethod
    def from_device(cls, ctx, d_surfels, n, d_tangent_mask=None):
        """tm_scene_create_dev: the cloud (and optionally tangent_mask_) are device buffers of ctx (ctx.alloc)"""
        self = cls.__new__(cls)
        self.ctx, self.n = ctx, int(n)
        h = vp()
        _check(lib().tm_scene_create_dev(ctx.h, d_surfels, self.n, d_tangent_mask, C.byref(h)))
        self.h = h
        return self

    def close(self):
        if getattr(self, "h", None):
            lib().tm_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def project(self, kind, state):
        """Device-resident uvw scene of a projector (tm_scene_create_projected); indices are this scene's."""
        st, pst = _f(state)
        h = vp()
        _check(lib().tm_scene_create_projected(self.h, kind, pst, C.byref(h)))
        out = Scene.__new__(Scene)
        out.ctx, out.n, out.h = self.ctx, self.n, h
        return out

    def probe_projected(self, kind, state, band=-1.0):
        """(points that project and are not masked, tangent points among them within `band` of the surface)
        without building the projected scene (tm_scene_probe_projected)"""
        st, pst = _f(state)
        live, tan = C.c_uint64(0), C.c_uint32(0)
        _check(lib().tm_scene_probe_projected(self.h, kind, pst, band, C.byref(live), C.byref(tan)))
        return live.value, tan.value

    def live_count(self):
        return int(lib().tm_scene_live_count(self.h))

    def download(self):
        s = np.zeros((self.n, 12), np.float32)
        tm = np.zeros(self.n, np.int32)
        _check(lib().tm_scene_download(self.h, s.ctypes.data_as(f32p), tm.ctypes.data_as(i32p)))
        return s, tm

    def tangent_points(self):
        """(ascending indices, surfels) of the points with tangent_mask set, as the kernels see them"""
        n = C.c_uint32(0)
        _check(lib().tm_scene_tangent_points(self.h, None, None, 0, C.byref(n)))
        idx = np.zeros(n.value, np.uint32)
        rows = np.zeros((n.value, 12), np.float32)
        if n.value:
            _check(lib().tm_scene_tangent_points(self.h, idx.ctypes.data_as(u32p), rows.ctypes.data_as(f32p), n.value,
                                                 C.byref(n)))
        return idx, rows

    def set_mask(self, mask):
        if mask is None:
            _check(lib().tm_scene_set_mask(self.h, None))
        else:
            m, p = _i(mask)
            _check(lib().tm_scene_set_mask(self.h, p))

    def mask_add(self, indices):
        """mask_[idx] = 1 for the listed points, on top of the current mask (reference include/impl/scene.hpp:87-90)"""
        idx, p = _u(indices)
        _check(lib().tm_scene_mask_add(self.h, p, idx.size))

    def curvature_subset(self, indices, k=30):
        """curvature(k, idx) of the listed points from the scene's device copy (tm_scene_curvature_subset)"""
        idx, pi = _u(indices)
        out = np.zeros((len(idx), 5), np.float32)
        _check(lib().tm_scene_curvature_subset(self.h, k, pi, len(idx), out.ctypes.data_as(f32p)))
        return out

    def set_tangent_mask(self, tangent_mask):
        m, p = _i(tangent_mask)
        _check(lib().tm_scene_set_tangent_mask(self.h, p))

    def auto_tangent_mask(self, k=30, min_norm=0.7, max_ratio=0.2, ok=None):
        """tm_scene_auto_tangent_mask: tangent_mask_ of the reference built on the device; returns
        (tangent_mask int32[n], candidates uint32[c], curv5 float32[c, 5]) and makes the mask the scene's"""
        okp = None
        if ok is not None:
            ok = np.ascontiguousarray(ok, dtype=np.uint8)
            okp = ok.ctypes.data_as(C.c_void_p)
        mask = np.zeros(self.n, np.int32)
        nc, nt = C.c_uint32(0), C.c_uint32(0)
        cand = np.zeros(self.n, np.uint32)
        rows = np.zeros((self.n, 5), np.float32)
        _check(lib().tm_scene_auto_tangent_mask(self.h, k, min_norm, max_ratio, okp, mask.ctypes.data_as(i32p),
                                                cand.ctypes.data_as(u32p), rows.ctypes.data_as(f32p), self.n,
                                                C.byref(nc), C.byref(nt)))
        assert nt.value == int(mask.sum())
        return mask, cand[:nc.value].copy(), rows[:nc.value].copy()

    def radius_subset(self, center_idx, radius):
        cnt = C.c_uint32(0)
        _check(lib().tm_scene_radius_subset(self.h, center_idx, radius, None, 0, C.byref(cnt)))
        out = np.zeros(cnt.value, np.int32)
        _check(lib().tm_scene_radius_subset(self.h, center_idx, radius, out.ctypes.data_as(i32p), out.size,
                                            C.byref(cnt)))
        return out

    def radius_count_and_select(self, center_idx, radius, ranks):
        """size of the neighbourhood (no list built) and its entries at the given ranks"""
        n = C.c_uint32(0)
        _check(lib().tm_scene_radius_subset(self.h, center_idx, radius, None, 0, C.byref(n)))
        r, pr = _u(ranks)
        out = np.zeros(len(r), np.int32)
        if len(r):
            _check(lib().tm_scene_radius_select(self.h, pr, len(r), out.ctypes.data_as(i32p)))
        return n.value, out

    def radius_lists(self, centers, radius):
        cn, pc = _u(centers)
        counts = np.zeros(cn.size, np.uint32)
        _check(lib().tm_scene_radius_lists(self.h, pc, cn.size, radius, None, 0, counts.ctypes.data_as(u32p)))
        stride = max(int(counts.max()) if cn.size else 0, 1)
        lists = np.zeros((cn.size, stride), np.int32)
        _check(lib().tm_scene_radius_lists(self.h, pc, cn.size, radius, lists.ctypes.data_as(i32p), stride,
                                           counts.ctypes.data_as(u32p)))
        return [lists[k, :counts[k]].copy() for k in range(cn.size)]

    def radius_tangent_lists(self, centers, radius):
        """(sizes of the neighbourhoods, per centre the (rank, index) pairs of its tangent members, ascending)"""
        c, pc = _u(centers)
        counts = np.zeros(len(c), np.uint32)
        tc = np.zeros(len(c), np.uint32)
        n = C.c_uint32(0)
        rc = lib().tm_scene_radius_tangent_lists(self.h, pc, len(c), radius, counts.ctypes.data_as(u32p),
                                                 tc.ctypes.data_as(u32p), None, 0, C.byref(n))
        if rc not in (OK, ERR_CAPACITY):
            _check(rc)
        pairs = np.zeros((max(n.value, 1), 2), np.uint32)
        _check(lib().tm_scene_radius_tangent_lists(self.h, pc, len(c), radius, counts.ctypes.data_as(u32p),
                                                   tc.ctypes.data_as(u32p), pairs.ctypes.data_as(u32p), n.value, C.byref(n)))
        offs = np.concatenate([[0], np.cumsum(tc)]).astype(np.int64)
        return counts, [pairs[offs[k]:offs[k + 1]].copy() for k in range(len(c))]

    def generate_hypotheses(self, model, ij_pairs, min_diameter_factor=0.2,
                            max_diameter_factor=1.0, force_up=False, query_limit=200, cap=None):
        ij, pij = _u(np.asarray(ij_pairs).reshape(-1, 2))
        n_pairs = ij.shape[0]
        if cap is None:
            lim = query_limit if query_limit > 0 else 4096
            cap = max(1, n_pairs * lim)
        poses = np.zeros((cap, 16), np.float32)
        prov = np.zeros((cap, 4), np.uint32)
        cnt, tot = C.c_uint32(0), C.c_uint32(0)
        _check(lib().tm_generate_hypotheses(self.h, model.h, pij, n_pairs, min_diameter_factor,
                                            max_diameter_factor, int(force_up), query_limit,
                                            poses.ctypes.data_as(f32p), prov.ctypes.data_as(u32p),
                                            cap, C.byref(cnt), C.byref(tot)))
        return poses[:cnt.value].copy(), prov[:cnt.value].copy(), tot.value

    def generate_hypotheses_dev(self, model, d_ij, n_pairs, d_poses, d_prov, cap,
                                min_diameter_factor=0.2, max_diameter_factor=1.0, force_up=False,
                                query_limit=200):
        cnt, tot = C.c_uint32(0), C.c_uint32(0)
        _check(lib().tm_generate_hypotheses_dev(self.h, model.h, d_ij, n_pairs, min_diameter_factor,
                                                max_diameter_factor, int(force_up), query_limit,
                                                d_poses, d_prov, cap, C.byref(cnt), C.byref(tot)))
        return cnt.value, tot.value

    def score_hypotheses(self, model, poses16, dist_thres, subset=None, mode=SCORE_AUTO,
                         traits=IDENTITY):
        P, pP = _f(np.asarray(poses16).reshape(-1, 16))
        n = P.shape[0]
        counts = np.zeros(n, np.uint32)
        scores = np.zeros(n, np.float64)
        if subset is None:
            ps, ns = None, 0
        else:
            subset, ps = _i(subset)
            ns = subset.size
        _check(lib().tm_score_hypotheses(self.h, model.h, traits, pP, n, ps, ns, dist_thres, mode,
                                         counts.ctypes.data_as(u32p), scores.ctypes.data_as(f64p)))
        return counts, scores

    def score_hypotheses_dev(self, model, d_poses, n_hyp, dist_thres, d_counts, d_scores,
                             mode=SCORE_AUTO, traits=IDENTITY):
        _check(lib().tm_score_hypotheses_dev(self.h, model.h, traits, d_poses, n_hyp, dist_thres,
                                             mode, d_counts, d_scores))

    def score_hypotheses_ball(self, model, poses16, centers, radius, dist_thres):
        P, pP = _f(np.asarray(poses16).reshape(-1, 16))
        c, pc = _u(centers)
        n = P.shape[0]
        counts = np.zeros(n, np.uint32)
        scores = np.zeros(n, np.float64)
        _check(lib().tm_score_hypotheses_ball(self.h, model.h, pP, n, pc, radius, dist_thres,
                                              counts.ctypes.data_as(u32p), scores.ctypes.data_as(f64p)))
        return counts, scores

    def score_hypotheses_on_points(self, model, poses16, points, dist_thres):
        P, pP = _f(np.asarray(poses16).reshape(-1, 16))
        pts, pp = _i(points)
        n = P.shape[0]
        counts = np.zeros(n, np.uint32)
        scores = np.zeros(n, np.float64)
        _check(lib().tm_score_hypotheses_on_points(self.h, model.h, pP, n, pp, pts.size, dist_thres,
                                                   counts.ctypes.data_as(u32p), scores.ctypes.data_as(f64p)))
        return counts, scores

    def score_hypotheses_early_drop(self, model, poses16, dist_thres, accept_prob, subset=None):
        P, pP = _f(np.asarray(poses16).reshape(-1, 16))
        n = P.shape[0]
        counts = np.zeros(n, np.uint32)
        scores = np.zeros(n, np.float64)
        dropped = np.zeros(n, np.int32)
        saved = np.zeros(n, np.uint32)
        if subset is None:
            ps, ns = None, 0
        else:
            subset, ps = _i(subset)
            ns = subset.size
        _check(lib().tm_score_hypotheses_early_drop(
            self.h, model.h, pP, n, ps, ns, dist_thres, accept_prob, counts.ctypes.data_as(u32p),
            scores.ctypes.data_as(f64p), dropped.ctypes.data_as(i32p), saved.ctypes.data_as(u32p)))
        return counts, scores, dropped, saved

    def score_hypotheses_early_drop_ball(self, model, poses16, centers, radius, dist_thres, accept_prob):
        P, pP = _f(np.asarray(poses16).reshape(-1, 16))
        cn, pc = _u(centers)
        n = P.shape[0]
        counts = np.zeros(n, np.uint32)
        scores = np.zeros(n, np.float64)
        dropped = np.zeros(n, np.int32)
        saved = np.zeros(n, np.uint32)
        _check(lib().tm_score_hypotheses_early_drop_ball(self.h, model.h, pP, n, pc, radius, dist_thres, accept_prob,
                                                         counts.ctypes.data_as(u32p), scores.ctypes.data_as(f64p),
                                                         dropped.ctypes.data_as(i32p), saved.ctypes.data_as(u32p)))
        return counts, scores, dropped, saved

    def score_hypotheses_early_drop_ball_dev(self, model, d_poses, n_hyp, centers, radius, dist_thres, accept_prob,
                                             d_counts, d_scores, d_dropped, d_saved):
        c, pc = _u(centers)
        _check(lib().tm_score_hypotheses_early_drop_ball_dev(self.h, model.h, d_poses, n_hyp, pc, radius, dist_thres,
                                                             accept_prob, d_counts, d_scores, d_dropped, d_saved))

    def search_batch(self, model, ij_pairs, radius, dist_thres, accept_prob, early_drop, sufficient,
                     min_diameter_factor=0.0, max_diameter_factor=1.0, force_up=False, query_limit=200):
        """One batch of find_in_subset on the device: (index or -1, n_hypotheses, count, dropped_at, score, pose16)"""
        ij, pij = _u(np.asarray(ij_pairs).reshape(-1))
        r = BatchResult()
        _check(lib().tm_search_batch(self.h, model.h, pij, ij.size // 2, min_diameter_factor, max_diameter_factor,
                                     1 if force_up else 0, query_limit, radius, dist_thres, accept_prob,
                                     1 if early_drop else 0, sufficient, C.byref(r)))
        return r.index, r.n_hypotheses, r.count, r.dropped_at, r.score, np.array(r.pose, np.float32)

    def search_batches(self, model, ij_pairs, batch_offsets, radius, dist_thres, accept_prob, early_drop, sufficient,
                       min_diameter_factor=0.0, max_diameter_factor=1.0, force_up=False, query_limit=200):
        """Several independent batches in one call: list of (index or -1, n_hypotheses, count, dropped_at, score,
        pose16) per batch"""
        ij, pij = _u(np.asarray(ij_pairs).reshape(-1))
        off, poff = _u(batch_offsets)
        nb = off.size - 1
        res = (BatchResult * nb)()
        _check(lib().tm_search_batches(self.h, model.h, pij, ij.size // 2, poff, nb, min_diameter_factor,
                                       max_diameter_factor, 1 if force_up else 0, query_limit, radius, dist_thres,
                                       accept_prob, 1 if early_drop else 0, sufficient, res))
        return [(r.index, r.n_hypotheses, r.count, r.dropped_at, r.score, np.array(r.pose, np.float32)) for r in res]

    def icp_refine(self, model, pose16, dist_thres, max_iterations):
        """icp_ on the device: (pose16, count, iterations)"""
        T, pT = _f(pose16)
        out = np.zeros(16, np.float32)
        cnt, it = C.c_uint32(0), C.c_uint32(0)
        _check(lib().tm_icp_refine(self.h, model.h, pT, dist_thres, max_iterations, out.ctypes.data_as(f32p),
                                   C.byref(cnt), C.byref(it)))
        return out, cnt.value, it.value

    def icp_refine_batch(self, model, poses16, dist_thres, max_iterations):
        P, pP = _f(np.asarray(poses16).reshape(-1, 16))
        n = P.shape[0]
        out = np.zeros((n, 16), np.float32)
        cnt = np.zeros(n, np.uint32)
        it = np.zeros(n, np.uint32)
        _check(lib().tm_icp_refine_batch(self.h, model.h, pP, n, dist_thres, max_iterations, out.ctypes.data_as(f32p),
                                         cnt.ctypes.data_as(u32p), it.ctypes.data_as(u32p)))
        return out, cnt, it

    def icp_accumulate(self, model, pose16, dist_thres):
        T, pT = _f(pose16)
        cnt = C.c_uint32(0)
        sums = np.zeros(15, np.int64)
        _check(lib().tm_icp_accumulate(self.h, model.h, pT, dist_thres, C.byref(cnt),
                                       sums.ctypes.data_as(C.POINTER(C.c_longlong))))
        return cnt.value, sums

    def extract_correspondences_batch(self, model, poses16, dist_thres):
        """[(scene_corrs, model_corrs, score)] per pose, one call"""
        P, pP = _f(np.asarray(poses16).reshape(-1, 16))
        n = P.shape[0]
        counts = np.zeros(n, np.uint32)
        scores = np.zeros(n, np.float64)
        rc = lib().tm_extract_correspondences_batch(self.h, model.h, pP, n, dist_thres, None, None, 0,
                                                    counts.ctypes.data_as(u32p), scores.ctypes.data_as(f64p))
        _check(rc)
        stride = max(int(counts.max()), 1)
        sc = np.zeros((n, stride), np.uint32)
        mc = np.zeros((n, stride), np.uint32)
        _check(lib().tm_extract_correspondences_batch(self.h, model.h, pP, n, dist_thres, sc.ctypes.data_as(u32p),
                                                      mc.ctypes.data_as(u32p), stride, counts.ctypes.data_as(u32p),
                                                      scores.ctypes.data_as(f64p)))
        return [(sc[k, :counts[k]].copy(), mc[k, :counts[k]].copy(), float(scores[k])) for k in range(n)]

    def extract_arbitrate_batch(self, model, poses16, dist_thres, min_points, min_points_all):
        """(verdict, kept, tangent_kept, [(scene_corrs, model_corrs, score)] per pose) — lists of rejected candidates
        are empty unless the verdict is -1 for all (the host decides)"""
        P, pP = _f(np.asarray(poses16).reshape(-1, 16))
        n = P.shape[0]
        counts = np.zeros(n, np.uint32)
        scores = np.zeros(n, np.float64)
        _check(lib().tm_extract_correspondences_batch(self.h, model.h, pP, n, dist_thres, None, None, 0,
                                                      counts.ctypes.data_as(u32p), scores.ctypes.data_as(f64p)))
        stride = max(int(counts.max()), 1)
        sc = np.full((n, stride), 0xFFFFFFFF, np.uint32)
        mc = np.full((n, stride), 0xFFFFFFFF, np.uint32)
        verdict = np.zeros(n, np.int32)
        kept = np.zeros(n, np.uint32)
        tan = np.zeros(n, np.uint32)
        _check(lib().tm_extract_arbitrate_batch(self.h, model.h, pP, n, dist_thres, int(min_points), int(min_points_all),
                                                sc.ctypes.data_as(u32p), mc.ctypes.data_as(u32p), stride,
                                                counts.ctypes.data_as(u32p), scores.ctypes.data_as(f64p),
                                                verdict.ctypes.data_as(i32p), kept.ctypes.data_as(u32p),
                                                tan.ctypes.data_as(u32p)))
        lists = []
        for k in range(n):
            if verdict[k] == 0:
                assert (sc[k] == 0xFFFFFFFF).all()  # a rejected candidate's rows are not written
                lists.append((np.zeros(0, np.uint32), np.zeros(0, np.uint32), float(scores[k])))
            else:
                lists.append((sc[k, :counts[k]].copy(), mc[k, :counts[k]].copy(), float(scores[k])))
        return verdict, kept, tan, counts, lists

    def mask_add_extracted(self, rows):
        r = np.ascontiguousarray(rows, np.uint32)
        _check(lib().tm_scene_mask_add_extracted(self.h, r.ctypes.data_as(u32p), r.size))

    def extract_correspondences(self, model, pose16, dist_thres):
        T, pT = _f(pose16)
        sc = np.zeros(self.n, np.uint32)
        mc = np.zeros(self.n, np.uint32)
        cnt = C.c_uint32(0)
        score = C.c_double(0)
        _check(lib().tm_extract_correspondences(self.h, model.h, pT, dist_thres,
                                                sc.ctypes.data_as(u32p), mc.ctypes.data_as(u32p),
                                                self.n, C.byref(cnt), C.byref(score)))
        return sc[:cnt.value].copy(), mc[:cnt.value].copy(), score.value


# ---- traits (host scalar) ----------------------------------------------------------------------
def traits_init_from_samples(kind, model_state, samples):
    ms, pm = _f(model_state)
    s, ps = _f(samples)
    st = np.zeros(STATE_FLOATS, np.float32)
    rej = lib().tm_traits_init_from_samples(kind, pm, ps, st.ctypes.data_as(f32p))
    return None if rej else st


def traits_project(kind, state, xyz):
    s, ps = _f(state)
    x, px = _f(xyz)
    out = np.zeros(3, np.float32)
    ok = lib().tm_traits_project(kind, ps, px, out.ctypes.data_as(f32p))
    return out if ok else None


def traits_unproject(kind, state, uvw):
    s, ps = _f(state)
    x, px = _f(uvw)
    out = np.zeros(3, np.float32)
    lib().tm_traits_unproject(kind, ps, px, out.ctypes.data_as(f32p))
    return out


def traits_tangent(kind, state, surfel):
    s, ps = _f(state)
    x, px = _f(surfel)
    out = np.zeros(3, np.float32)
    lib().tm_traits_tangent(kind, ps, px, out.ctypes.data_as(f32p))
    return out


def traits_normal(kind, state, surfel):
    s, ps = _f(state)
    x, px = _f(surfel)
    out = np.zeros(3, np.float32)
    lib().tm_traits_normal(kind, ps, px, out.ctypes.data_as(f32p))
    return out


def traits_intrinsic_distance(kind, state, p0, p1):
    s, ps = _f(state)
    a, pa = _f(p0)
    b, pb = _f(p1)
    return float(lib().tm_traits_intrinsic_distance(kind, ps, pa, pb))


def rigid_fit(sums15, count):
    """tm_rigid_fit: the ICP loop's rigid fit (Horn) from the 15 fixed-point sums, on the host"""
    v = np.ascontiguousarray(sums15, dtype=np.int64)
    out = np.zeros(16, np.float32)
    _check(lib().tm_rigid_fit(v.ctypes.data_as(C.POINTER(C.c_longlong)), int(count), out.ctypes.data_as(f32p)))
    return out
