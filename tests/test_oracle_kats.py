"""Pins the CPU oracle (oracle/) against the known-answer values of SURVEY.md Appendix C and
against structural properties — the reference itself ships no tests or fixtures (parity unpinned)."""
import numpy as np
import pytest

from oracle import tmo
from triplet_match_amd import synth

STEP = float(np.float32(0.17453294))


def test_murmur3_public_seed0_vectors():
    # public MurmurHash3_x86_32 vectors (seed 0), little-endian u32 words
    assert tmo.murmur3_32([0x00000000], 0) == 0x2362F9DE
    assert tmo.murmur3_32([0xFFFFFFFF], 0) == 0x76293B50
    assert tmo.murmur3_32([0x87654321], 0) == 0xF55B516B


@pytest.mark.parametrize("key,expect", [
    ([0, 0, 0, 0], 0xEE18D2A5), ([1, 2, 3, 4], 0x3F7F5D44), ([3, 9, 9, 3], 0x29DA4D9D),
    ([19, 0, 17, 19], 0xC8D595AD), ([0xFFFFFFFF] * 4, 0xB7DBE945), ([7], 0x4054B70B),
    ([5, 6, 7], 0xF8B3752F)])
def test_murmur_seed42(key, expect):
    # reference include/impl/discretize.hpp:10 murmur_seed = 42
    assert tmo.murmur_key(key) == expect


def test_discretize_kats():
    assert tmo.discretize_range(0.5, 0, 1, 20) == 10
    assert tmo.discretize_range(1.0, 0, 1, 20) == 19
    assert tmo.discretize_range(-0.1, 0, 1, 20) == 0
    assert tmo.discretize_range(0.999, 0, 1, 20) == 19
    assert tmo.discretize_range(0.05, 0, 1, 20) == 1
    assert tmo.discretize_step(0.35, STEP) == 2
    assert tmo.discretize_step(float(np.float32(np.pi / 2)), STEP) == 9  # exact bin edge
    assert tmo.discretize_step(float(np.nextafter(np.float32(np.pi / 2), np.float32(0))), STEP) == 8
    # stale-app quirk: steps = 0 -> steps - 1 wraps
    assert tmo.discretize_range(2.0, 0, 1, 0) == 0xFFFFFFFF
    # zero range -> NaN nval -> 0
    assert tmo.discretize_range(1.0, 1.0, 0.0, 20) == 0


def test_feature_closed_form():
    s0 = np.zeros(12, np.float32)
    s1 = np.zeros(12, np.float32)
    s1[0] = 1
    s0[10] = 1  # t0 = (0,1,0)
    s1[9] = 1   # t1 = (1,0,0)
    f = tmo.feature(s0, s1)
    assert f[0] == 1.0 and f[3] == 1.0
    assert f[1] == np.float32(np.pi / 2)
    assert f[2] == 0.0
    s0[9:12] = np.array([1, 1, 0], np.float32) / np.float32(np.sqrt(2))
    assert abs(tmo.feature(s0, s1)[1] - np.pi / 4) < 2e-7


def test_atan2_matches_libm_within_2ulp():
    rng = np.random.default_rng(0)
    y = rng.normal(size=20000).astype(np.float32)
    x = rng.normal(size=20000).astype(np.float32)
    mine = np.array([tmo.atan2f(float(a), float(b)) for a, b in zip(y, x)], dtype=np.float32)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.max(np.abs(mine - ref)) < 2 * np.spacing(np.float32(np.pi))
    assert tmo.atan2f(1.0, 0.0) == float(np.float32(np.pi / 2))
    assert tmo.atan2f(0.0, 1.0) == 0.0
    assert tmo.atan2f(1.0, 1.0) == float(np.float32(np.pi / 4))


def test_early_drop_bound_kats():
    # (100,1000,10) was checked against the literal expression compiled with g++
    assert tmo.early_drop_upper(100, 1000, 10) == 136
    assert tmo.early_drop_upper(50, 1000, 0) == 36
    assert tmo.early_drop_upper(500, 1000, 400) == 812
    rng = np.random.default_rng(1)
    for _ in range(2000):
        S = int(rng.integers(1, 2_000_000))
        tried = int(rng.integers(1, S + 1))
        corrs = int(rng.integers(0, tried + 1))
        assert tmo.early_drop_upper(tried, S, corrs) == tmo.early_drop_upper_literal(tried, S, corrs)
    cp = tmo.early_drop_checkpoints(1000)
    assert list(cp[:3]) == [50, 100, 150] and cp[17] == 900 and len(cp) == 18


def test_base_transform_identity_and_rigid():
    rng = np.random.default_rng(2)
    a, b = rng.normal(size=3), rng.normal(size=3)
    t = np.cross(b - a, rng.normal(size=3))
    t /= np.linalg.norm(t)
    T = tmo.base_transform(a, b, t, a, b, t).reshape(4, 4).T
    assert np.allclose(T, np.eye(4), atol=2e-6)
    R = synth.random_rotation(rng)
    tr = rng.normal(size=3)
    T = tmo.base_transform(a, b, t, R @ a + tr, R @ b + tr, R @ t).reshape(4, 4).T
    assert np.allclose(T[:3, :3], R, atol=5e-6) and np.allclose(T[:3, 3], tr, atol=2e-5)
    assert np.all(T[3] == [0, 0, 0, 1])


def test_voxel_build_kdtree_equals_bruteforce():
    model, _ = synth.synth_model(300, seed=3)
    res = synth.resolution(model)
    a = tmo.Model(model, res)
    a.build_voxels(threads=4)
    b = tmo.Model(model, res)
    b.build_voxels(threads=4, brute=True)
    assert np.array_equal(a.voxels(), b.voxels())
    ext, tv = a.grid()
    assert a.voxels().size == int(ext[0]) * int(ext[1]) * int(ext[2])
    # voxel_query: truncation toward zero maps (-1,0) into cell 0
    inv = lambda v, d: (v - tv[12 + d]) / tv[d * 5]
    p = [inv(-0.5, 0), inv(0.5, 1), inv(0.5, 2), 1.0]
    assert a.voxel_query(p) == a.voxels()[0]
    p = [inv(-1.5, 0), inv(0.5, 1), inv(0.5, 2), 1.0]
    assert a.voxel_query(p) is None
    p = [inv(float(ext[0]) + 0.5, 0), inv(0.5, 1), inv(0.5, 2), 1.0]
    assert a.voxel_query(p) is None


def test_voxel_grid_against_scipy():
    """a6 pinned independently of both of the oracle's searches: the nearest model point of every voxel's query
    position inv(to_voxel) * (i, j, k, 1) (reference include/impl/model.hpp:60-84; the matrix inverted here by numpy
    in float64) from scipy's cKDTree. The same point wherever the nearest neighbour is unique by a margin, the same
    distance (to float32 rounding of the query position) everywhere."""
    from scipy.spatial import cKDTree
    model, _ = synth.synth_model(2500, seed=11)
    res = synth.resolution(model)
    m = tmo.Model(model, res)
    m.build_voxels(threads=8)
    ext, tv = m.grid()
    T = tv.reshape(4, 4).T.astype(np.float64)  # column-major 4x4
    inv = np.linalg.inv(T)
    k, j, i = np.meshgrid(np.arange(ext[2]), np.arange(ext[1]), np.arange(ext[0]), indexing="ij")  # lin = k*ex*ey + j*ex + i
    ijk1 = np.stack([i.ravel(), j.ravel(), k.ravel(), np.ones(i.size)], axis=1).astype(np.float64)
    q = (ijk1 @ inv.T)[:, :3]
    pts = model[:, :3].astype(np.float64)
    d, idx = cKDTree(pts).query(q, k=2)
    vox = m.voxels()
    assert vox.size == q.shape[0] > 100000
    d_oracle = np.linalg.norm(pts[vox] - q, axis=1)
    scale = np.abs(pts).max() + np.abs(q).max()
    assert np.all(np.abs(d_oracle - d[:, 0]) <= 4e-6 * scale)  # the same nearest distance everywhere
    clear = (d[:, 1] - d[:, 0]) > 2e-5 * scale                 # ... and the same point where it is the nearest by a margin
    assert clear.mean() > 0.9
    assert np.array_equal(vox[clear], idx[clear, 0])


def test_query_before_init_raises():
    model, _ = synth.synth_model(200, seed=3)
    m = tmo.Model(model, synth.resolution(model))
    with pytest.raises(RuntimeError, match="uninitialized"):
        m.query([1, 0, 0, 1])


def test_table_is_reverse_insertion_order(oracle_small, small_case):
    """libstdc++ equal_range order = reverse insertion: (i,j) positions descending."""
    m = oracle_small["model"]
    keys, counts = m.table_keys()
    assert counts.sum() == m.pair_count()
    subset = small_case["subset"]
    rank = {int(v): k for k, v in enumerate(subset)}
    big = keys[np.argmax(counts)]
    entries, total = m.query_key(big)
    assert total == counts.max()
    order = [rank[int(i)] * len(subset) + rank[int(j)] for i, j in entries]
    assert order == sorted(order, reverse=True)
    assert np.all(keys[:, 0] == keys[:, 3])  # k3 == k0
    assert keys[:, 0].max() <= 19 and keys[:, 1].max() <= 9 and keys[:, 2].max() <= 9


def test_planted_pose_scores_high(oracle_small, small_case):
    m, s = oracle_small["model"], oracle_small["scene"]
    for T in small_case["scene"]["poses_scene_to_model"]:
        sc, mc, score, saved = s.project(m, synth.pose_colmajor(T), 0.5, 3.0)
        assert len(sc) > 0.85 * 2000 and 0.8 < score <= 1.0 and saved == 0
        assert np.all(np.diff(sc.astype(np.int64)) > 0)  # scan order = ascending index
    sc, _, _, _ = s.project(m, synth.pose_colmajor(np.eye(4)), 0.5, 3.0)
    assert len(sc) < 100


def test_project_early_drop_semantics(oracle_small, small_case):
    m, s = oracle_small["model"], oracle_small["scene"]
    T = synth.pose_colmajor(np.eye(4))
    sc_full, _, score_full, _ = s.project(m, T, 0.5, 3.0)
    sc, _, score, saved = s.project(m, T, 0.5, 3.0, early_out=True)
    # a far-off pose never reaches a voxel hit -> the test at the loop tail never runs
    assert saved == 0 or len(sc) <= len(sc_full)
    Tgood = synth.pose_colmajor(small_case["scene"]["poses_scene_to_model"][0])
    sub = s.radius_subset(int(np.nonzero(small_case["scene"]["instance_id"] == 0)[0][0]), m.diameter)
    a = s.project(m, Tgood, 0.5, 3.0, subset=sub)
    b = s.project(m, Tgood, 0.5, 3.0, subset=sub, early_out=True)
    assert b[3] == 0 and np.array_equal(a[0], b[0])  # a good pose survives every checkpoint
    # a slightly-off pose gets dropped and reports the raw (un-normalised) partial score
    Tbad = np.array(small_case["scene"]["poses_scene_to_model"][0])
    Tbad[:3, 3] += 0.35
    c = s.project(m, synth.pose_colmajor(Tbad), 0.5, 3.0, subset=sub, early_out=True)
    assert c[3] > 0 and len(c[0]) < len(a[0])


def test_mask_removes_points(oracle_small, small_case):
    m, s = oracle_small["model"], oracle_small["scene"]
    T = synth.pose_colmajor(small_case["scene"]["poses_scene_to_model"][0])
    sc, _, _, _ = s.project(m, T, 0.5, 3.0)
    mask = np.zeros(s.n, np.int32)
    mask[sc[::2]] = 1
    s.set_mask(mask)
    sc2, _, _, _ = s.project(m, T, 0.5, 3.0)
    s.set_mask(None)
    assert np.array_equal(sc2, sc[1::2])


def alignment_error(case, T):
    """mean distance between T*p and T_true*p over the points of the best-matching instance"""
    sc = case["scene"]
    best = np.inf
    for k, Tt in enumerate(sc["poses_scene_to_model"]):
        p = sc["surfels"][sc["instance_id"] == k, 0:3].astype(np.float64)
        a = p @ T[:3, :3].T + T[:3, 3]
        b = p @ Tt[:3, :3].T + Tt[:3, 3]
        best = min(best, float(np.linalg.norm(a - b, axis=1).mean()))
    return best


def test_generate_hypotheses_recovers_pose(oracle_small, small_case):
    from conftest import sample_pairs
    m, s = oracle_small["model"], oracle_small["scene"]
    pairs = sample_pairs(small_case, m)
    poses, prov = s.generate_hypotheses(m, pairs)
    assert len(poses) > 100 and prov[:, 3].max() < 200
    assert np.all(np.diff(prov[:, 0].astype(np.int64)) >= 0)  # visit order
    counts, scores, ev = s.score_hypotheses(m, poses, 0.5, 3.0, threads=4)
    best = poses[np.argmax(counts)].reshape(4, 4).T.astype(np.float64)
    err = alignment_error(small_case, best)
    assert counts.max() > 0.5 * 2000 and err < 5 * small_case["resolution"]
    assert ev == len(poses) * s.n


def test_traits_roundtrip_and_kats():
    rng = np.random.default_rng(4)
    ms = np.zeros(tmo.STATE_FLOATS, np.float32)
    ms[1] = 0.2  # threshold
    # cylinder of radius 0.5 about the z axis through (1,2,3)
    def cyl_surfel(theta, h, r=0.5):
        s = np.zeros(12, np.float32)
        n = np.array([np.cos(theta), np.sin(theta), 0.0])
        s[0:3] = np.array([1, 2, 3]) + r * n + np.array([0, 0, h])
        s[4:7] = -n  # inward normal: origin = p + n*(...) lands on the axis
        s[9:12] = [0, 0, 1]
        return s
    smp = np.concatenate([cyl_surfel(0.3, 0.1), cyl_surfel(1.7, 0.9)])
    st = tmo.traits_init_from_samples(tmo.CYLINDER, ms, smp)
    assert st is not None and abs(st[0] - 0.5) < 1e-5
    p = cyl_surfel(2.5, 0.4)
    uvw = tmo.traits_project(tmo.CYLINDER, st, p[0:3])
    assert uvw is not None and abs(uvw[2]) < 1e-5 and 0 <= uvw[0] <= 2 * np.pi * 0.5 + 1e-5
    back = tmo.traits_unproject(tmo.CYLINDER, st, uvw)
    assert np.allclose(back, p[0:3], atol=1e-5)
    far = p[0:3] + 0.5 * np.array([np.cos(2.5), np.sin(2.5), 0])
    assert tmo.traits_project(tmo.CYLINDER, st, far) is None
    # wrap-around distance
    d = tmo.traits_intrinsic_distance(tmo.CYLINDER, st, [0.01, 0, 0], [2 * np.pi * 0.5 - 0.01, 0, 0])
    assert abs(d - 0.02) < 1e-5
    # plane2 rejects inconsistent normals (returns nullptr in the reference)
    s3 = np.zeros((3, 12), np.float32)
    s3[:, 0:3] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    s3[:, 4:7] = [0, 0, 1]
    assert tmo.traits_init_from_samples(tmo.PLANE2, ms, s3.ravel()) is not None
    s3[1, 4:7] = [1, 0, 0]
    assert tmo.traits_init_from_samples(tmo.PLANE2, ms, s3.ravel()) is None
    # plane: threshold on |w|
    pl = np.zeros(12, np.float32)
    pl[4:7] = [0, 0, 1]
    st = tmo.traits_init_from_samples(tmo.PLANE, ms, pl)
    assert tmo.traits_project(tmo.PLANE, st, [0.3, 0.4, 0.1]) is not None
    assert tmo.traits_project(tmo.PLANE, st, [0.3, 0.4, 0.3]) is None
    assert abs(tmo.traits_intrinsic_distance(tmo.PLANE2, st, [0, 0, 0], [3, 4, 9]) - 5) < 1e-6


# ---- independent pins (vectors / counts that do not come from our own arithmetic) ----
def _golden(name):
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)


def test_curvature_and_resolution_against_scipy_and_numpy():
    """cloud->curvature(30, i) and cloud->resolution() of the oracle (own exhaustive kNN, f32 running mean of the
    projected normals, own Jacobi iteration) against tests/golden/curvature_20k.npz = scipy cKDTree + numpy.linalg.eigh
    in float64 (make_curvature_golden.py): same neighbour sets up to ties (equal 30th-neighbour distances), principal
    curvatures to 1e-6, nearest-neighbour distances to 1e-6 relative. The reference's mask `pc_min / pc_max < 0.2`
    then differs only where the ratio sits on the threshold itself."""
    import hashlib
    from triplet_match_amd import synth
    g = np.load(_golden("curvature_20k.npz"))
    cloud, _ = synth.synth_model(20000, seed=4)
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(cloud).tobytes()).digest(), np.uint8)
    assert np.array_equal(sha, g["cloud_sha256"]), "the generator no longer makes the cloud the vectors were taken on"
    c = tmo.cloud_curvature(cloud, int(g["k"]), threads=8)
    assert np.abs(c[:, 3] - g["pc_min"]).max() < 1e-6 and np.abs(c[:, 4] - g["pc_max"]).max() < 1e-6
    res, nn1 = tmo.cloud_resolution(cloud, threads=8)
    assert (np.abs(nn1 - g["nn1"]) <= 1e-6 * g["nn1"]).all()
    assert abs(res - float(np.mean(g["nn1"].astype(np.float64)))) < 1e-5 * res
    with np.errstate(invalid="ignore", divide="ignore"):
        mask_o = (c[:, 3] / c[:, 4]) < 0.2
        ratio_g = g["pc_min"].astype(np.float64) / g["pc_max"].astype(np.float64)
    differ = np.nonzero(mask_o != (ratio_g < 0.2))[0]
    assert len(differ) <= 5 and (np.abs(ratio_g[differ] - 0.2) < 1e-6).all()
    rng = np.random.default_rng(2)
    for i in rng.integers(0, len(cloud), 300):  # the oracle's kNN ends at the same distance as scipy's
        _, d2 = tmo.cloud_knn(cloud, cloud[i, 0:3], int(g["k"]))
        assert abs(d2[-1] - g["kth_d2"][i]) <= 2e-6 * g["kth_d2"][i] + 1e-12


def test_atan2_census_is_what_was_committed():
    """The reference's binary calls glibc atan2f, the project its own (include/impl/feature.hpp:7).
    tests/golden/atan2_census.json says how many descriptor keys that changes at the configs[1] size: 4 of the
    14,051,647 table entries, 0 of 869,958 scene pairs. Here: the file is consistent, and the small census in it is
    reproduced from scratch (same glibc, same generator)."""
    import json
    import importlib.util
    rec = json.load(open(_golden("atan2_census.json")))
    full = rec["full_size"]
    assert full["model_pairs"]["pairs"] == 14051647 and full["model_pairs"]["bins_differ"] == 4
    assert len(full["model_pairs"]["differing_ij"]) == full["model_pairs"]["bins_differ"]
    assert full["scene_pairs"]["bins_differ"] == 0 and full["scene_pairs"]["pairs"] > 800000
    assert full["model_pairs"]["max_abs_diff"] <= 2.4e-7  # <= 1 ulp at pi / 2
    spec = importlib.util.spec_from_file_location("make_atan2_census", _golden("make_atan2_census.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    small = mod.census(5_000, 50_000, 50_000)
    for part in ("model_pairs", "scene_pairs"):
        for key in ("pairs", "bins_differ", "bits_differ"):
            assert small[part][key] == rec["small"][part][key], (part, key)


def test_early_drop_bound_against_the_literal_expression_on_a_grid():
    """the restated bound (positive operands) == the literal transcription of include/impl/scene.hpp:493-498
    (negative operands, negative-double -> uint32 cast as x86-64 does it) on a grid of 3.3e8 (tried, subset size,
    correspondences) triples up to subsets of a million points — not three known answers; the product's host function
    agrees on a random sample of them"""
    n, bad, first = tmo.early_drop_compare(1_000_000, 9973, 499, 101)
    assert n > 300_000_000 and bad == 0, first
    n, bad, first = tmo.early_drop_compare(1200, 1, 1, 1)  # every case up to subsets of 1,200 points
    assert n > 200_000_000 and bad == 0, first
    from triplet_match_amd import capi
    rng = np.random.default_rng(7)
    for _ in range(20000):
        S = int(rng.integers(1, 1_000_001))
        tried = int(rng.integers(1, S + 1))
        corrs = int(rng.integers(0, tried + 1))
        assert capi.early_drop_upper(tried, S, corrs) == tmo.early_drop_upper_literal(tried, S, corrs), (tried, S, corrs)


def _project_numpy(model, resolution, voxels, extents, to_voxel, scene, tangent_mask, mask, pose16, dist_thres):
    """project_ of the reference (include/impl/scene.hpp:411-510, full pass) written a second time, in numpy, line by
    line from the reference text — not from the oracle's C++: float32 element-wise arithmetic in the association of
    SURVEY Q1 (a 4x4 times (x, y, z, 1) row = ((m0 x + m4 y) + m8 z) + m12), the voxel look-up of impl/model.hpp:180-192
    (int cast = truncation toward zero), the fp64 score sum. Returns (scene_corrs, model_corrs, score)."""
    f32 = np.float32
    T = np.asarray(pose16, f32).reshape(4, 4).T  # column-major storage -> T[row, col]
    P = scene[:, 0:3].astype(f32)
    x, y, z = P[:, 0], P[:, 1], P[:, 2]

    def row(r):  # :439-440: pos = T * p
        return ((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3]
    px, py, pz = row(0), row(1), row(2)
    V = np.asarray(to_voxel, f32).reshape(4, 4).T  # impl/model.hpp:181: to_voxel * pos (a scale and a translation)
    with np.errstate(invalid="ignore", over="ignore"):
        vx, vy, vz = V[0, 0] * px + V[0, 3], V[1, 1] * py + V[1, 3], V[2, 2] * pz + V[2, 3]
        fin = np.isfinite(vx) & np.isfinite(vy) & np.isfinite(vz)
        i = np.where(fin, np.trunc(np.where(fin, vx, 0)), -1).astype(np.int64)  # :182: cast<int>
        j = np.where(fin, np.trunc(np.where(fin, vy, 0)), -1).astype(np.int64)
        k = np.where(fin, np.trunc(np.where(fin, vz, 0)), -1).astype(np.int64)
    ex, ey, ez = (int(e) for e in extents)
    inside = (i >= 0) & (i < ex) & (j >= 0) & (j < ey) & (k >= 0) & (k < ez)  # :184-189
    live = inside & (np.asarray(mask) == 0)  # :434
    cand = np.nonzero(live)[0]
    m = voxels[(k[cand] * ey + j[cand]) * ex + i[cand]].astype(np.int64)  # :190: voxel_data_[k * ex * ey + j * ex + i]
    M = model[:, 0:3].astype(f32)
    dx, dy, dz = px[cand] - M[m, 0], py[cand] - M[m, 1], pz[cand] - M[m, 2]
    dist = np.sqrt((dx * dx + dy * dy) + dz * dz)  # :463 norm()
    thres = f32(dist_thres) * f32(resolution)      # :413
    keep = ~(dist > thres)                         # :464
    cand, m = cand[keep], m[keep]
    mt = model[m, 9:12].astype(f32)
    m_is_t = np.sqrt((mt[:, 0] * mt[:, 0] + mt[:, 1] * mt[:, 1]) + mt[:, 2] * mt[:, 2]) > f32(0.7)  # :469-470
    s_is_t = np.asarray(tangent_mask)[cand] != 0   # :441
    keep = m_is_t == s_is_t                        # :472-478
    cand, m, s_is_t = cand[keep], m[keep], s_is_t[keep]
    sv = np.where(s_is_t[:, None], scene[cand, 9:12], scene[cand, 4:7]).astype(f32)   # tangent or normal of p
    mv = np.where(s_is_t[:, None], model[m, 9:12], model[m, 4:7]).astype(f32)         # ... and of P_m
    rx = (T[0, 0] * sv[:, 0] + T[0, 1] * sv[:, 1]) + T[0, 2] * sv[:, 2]               # R * v, no translation
    ry = (T[1, 0] * sv[:, 0] + T[1, 1] * sv[:, 1]) + T[1, 2] * sv[:, 2]
    rz = (T[2, 0] * sv[:, 0] + T[2, 1] * sv[:, 1]) + T[2, 2] * sv[:, 2]
    term = np.abs((rx * mv[:, 0] + ry * mv[:, 1]) + rz * mv[:, 2])                   # :484
    score = float(np.sum(term.astype(np.float64))) / float(len(model))                # :486, :509
    return cand.astype(np.uint32), m.astype(np.uint32), score


def test_project_against_a_numpy_restatement(oracle_small, small_case):
    """The oracle's project_ against the reference text written a second time in another language: correspondence
    lists equal entry for entry, scores to the last few bits of the fp64 sum — for the planted poses, perturbed ones,
    random ones, with mask_ set and with non-finite points in the scene."""
    m, res = oracle_small["model"], small_case["resolution"]
    model = small_case["model"]
    S = small_case["scene"]["surfels"].copy()
    S[17, 0] = np.nan
    S[4242, 1] = np.inf
    tmk = small_case["scene"]["tangent_mask"]
    s = tmo.Scene(S, tmk)
    ext, tv = m.grid()
    vox = m.voxels()
    rng = np.random.default_rng(12)
    mask = (rng.random(len(S)) < 0.3).astype(np.int32)
    poses = [synth.pose_colmajor(T) for T in small_case["scene"]["poses_scene_to_model"]]
    for base in small_case["scene"]["poses_scene_to_model"]:
        for _ in range(3):
            T = np.array(base, np.float64)
            T[:3, 3] += rng.normal(0, 0.5 * res, 3)
            poses.append(synth.pose_colmajor(T))
    for _ in range(4):
        T = np.eye(4)
        T[:3, :3] = synth.random_rotation(rng)
        T[:3, 3] = rng.normal(size=3)
        poses.append(synth.pose_colmajor(T))
    hits = 0
    for use_mask in (False, True):
        s.set_mask(mask if use_mask else None)
        for dt in (3.0, 0.5):
            for pose in poses:
                sc, mc, score, saved = s.project(m, pose, 0.5, dt)
                nsc, nmc, nscore = _project_numpy(model, res, vox, ext, tv, S, tmk, mask if use_mask else np.zeros(len(S), np.int32),
                                                  pose, dt)
                assert np.array_equal(sc, nsc) and np.array_equal(mc, nmc), (use_mask, dt)
                assert abs(score - nscore) <= 1e-12 * max(1.0, abs(score)) and saved == 0
                hits += int(len(sc) > 500)
    assert hits >= 12


def test_pair_table_against_a_numpy_restatement(oracle_small, small_case):
    """The pair table (reference include/impl/model.hpp:100-149 with feature.hpp:16-46,50-88,91-114 and
    src/discretize.cpp:20-30) written a second time in numpy from the reference text: the same filters, the same
    bounds pass (incl. valid_bounds' min + 1.f * (max - min) in float32), the same keys with the same multiplicities.
    numpy's arctan2 stands in for atan2f (the census of tests/golden/atan2_census.json: a handful of 1.4e7 keys move):
    at most two entries may sit in a neighbouring angle bin."""
    f32 = np.float32
    m = oracle_small["model"]
    model, subset = small_case["model"], np.asarray(small_case["subset"], np.int64)
    P, Tg = model[:, 0:3].astype(f32), model[:, 9:12].astype(f32)
    diam = f32(m.diameter)
    lower, upper = diam * f32(0.2), diam * f32(1.0)
    I, J = np.meshgrid(subset, subset, indexing="ij")  # i-major, j-minor (:100)
    I, J = I.ravel(), J.ravel()
    ne = I != J
    I, J = I[ne], J[ne]
    d = P[J] - P[I]
    dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    with np.errstate(invalid="ignore", divide="ignore"):
        dn = d / dist[:, None]                                                      # :108 d1 /= dist1
        dott = (dn[:, 0] * Tg[I, 0] + dn[:, 1] * Tg[I, 1]) + dn[:, 2] * Tg[I, 2]
        keep = ~((dist < lower) | (dist > upper)) & ~((f32(1.0) - np.abs(dott)) < f32(0.01))  # :109-110
    I, J, d, dist = I[keep], J[keep], d[keep], dist[keep]

    def angle(a, b):  # feature.hpp:5-8
        cx = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
        cy = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
        cz = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
        cn = np.sqrt((cx * cx + cy * cy) + cz * cz)
        dt = np.abs((a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]) + a[:, 2] * b[:, 2])
        return np.arctan2(cn, dt).astype(f32)
    f0, f1, f2 = dist, angle(d, Tg[I]), angle(d, Tg[J])
    bmin0, bmax0 = f0.min(), f0.max()                                               # :115 extend
    diag0 = bmax0 - bmin0
    nmin0, nmax0 = bmin0 + f32(0.0) * diag0, bmin0 + f32(1.0) * diag0               # valid_bounds(…, 0.0f, 1.f), :122
    got_min, got_max = m.feature_bounds()
    assert got_min[0] == nmin0 and got_max[0] == nmax0
    pi = f32(np.pi)
    ok = ~((f0 < nmin0) | (f0 > nmax0)) & (f1 >= 0) & (f1 <= pi) & (f2 >= 0) & (f2 <= pi)  # valid(), feature.hpp:50-88
    rng0 = nmax0 - nmin0                                                            # bounds.diagonal()[0] of the NEW bounds
    steps = np.uint32(np.float32(20.0))                                             # float -> uint32_t parameter
    nval = (f0 - nmin0) / rng0
    k0 = np.where(nval < 0, 0, np.where(nval >= 1, steps - 1, (nval * f32(steps)).astype(np.uint32))).astype(np.uint32)
    step = f32(10.0) * f32(np.pi) / f32(180.0)
    k1, k2 = (f1 / step).astype(np.uint32), (f2 / step).astype(np.uint32)
    keys = np.stack([k0, k1, k2, k0], 1)[ok]
    want = {}
    for k in map(tuple, keys.tolist()):
        want[k] = want.get(k, 0) + 1
    okeys, ocounts = m.table_keys()
    got = {tuple(int(x) for x in k): int(c) for k, c in zip(okeys, ocounts)}
    assert sum(got.values()) == sum(want.values()) == int(ok.sum()) == m.pair_count() and len(keys) > 1000
    moved = sum(abs(got.get(k, 0) - want.get(k, 0)) for k in set(got) | set(want))
    assert moved <= 4, moved  # an entry that changes bin counts twice


def _point_outcomes_numpy(model, resolution, voxels, extents, to_voxel, scene, tangent_mask, pose16, dist_thres):
    """per scene point, from the reference text (as _project_numpy): inside the voxel grid?, correspondence?, its model
    partner, its score term |ref . ref_n| as float32"""
    f32 = np.float32
    T = np.asarray(pose16, f32).reshape(4, 4).T
    P = scene[:, 0:3].astype(f32)
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    px = ((T[0, 0] * x + T[0, 1] * y) + T[0, 2] * z) + T[0, 3]
    py = ((T[1, 0] * x + T[1, 1] * y) + T[1, 2] * z) + T[1, 3]
    pz = ((T[2, 0] * x + T[2, 1] * y) + T[2, 2] * z) + T[2, 3]
    V = np.asarray(to_voxel, f32).reshape(4, 4).T
    with np.errstate(invalid="ignore", over="ignore"):
        v = [V[0, 0] * px + V[0, 3], V[1, 1] * py + V[1, 3], V[2, 2] * pz + V[2, 3]]
        fin = np.isfinite(v[0]) & np.isfinite(v[1]) & np.isfinite(v[2])
        ijk = [np.where(fin, np.trunc(np.where(fin, c, 0)), -1).astype(np.int64) for c in v]
    ex, ey, ez = (int(e) for e in extents)
    inside = (ijk[0] >= 0) & (ijk[0] < ex) & (ijk[1] >= 0) & (ijk[1] < ey) & (ijk[2] >= 0) & (ijk[2] < ez)
    lin = np.where(inside, (ijk[2] * ey + ijk[1]) * ex + ijk[0], 0)
    m = voxels[lin].astype(np.int64)
    M = model[:, 0:3].astype(f32)
    with np.errstate(invalid="ignore", over="ignore"):
        dx, dy, dz = px - M[m, 0], py - M[m, 1], pz - M[m, 2]
        dist = np.sqrt((dx * dx + dy * dy) + dz * dz)
    close = inside & ~(dist > f32(dist_thres) * f32(resolution))
    mt = model[m, 9:12].astype(f32)
    m_is_t = np.sqrt((mt[:, 0] * mt[:, 0] + mt[:, 1] * mt[:, 1]) + mt[:, 2] * mt[:, 2]) > f32(0.7)
    s_is_t = np.asarray(tangent_mask) != 0
    corr = close & (m_is_t == s_is_t)
    sv = np.where(s_is_t[:, None], scene[:, 9:12], scene[:, 4:7]).astype(f32)
    mv = np.where(s_is_t[:, None], model[m, 9:12], model[m, 4:7]).astype(f32)
    with np.errstate(invalid="ignore", over="ignore"):
        rx = (T[0, 0] * sv[:, 0] + T[0, 1] * sv[:, 1]) + T[0, 2] * sv[:, 2]
        ry = (T[1, 0] * sv[:, 0] + T[1, 1] * sv[:, 1]) + T[1, 2] * sv[:, 2]
        rz = (T[2, 0] * sv[:, 0] + T[2, 1] * sv[:, 1]) + T[2, 2] * sv[:, 2]
        term = np.abs((rx * mv[:, 0] + ry * mv[:, 1]) + rz * mv[:, 2])
    return inside, corr, m, term


def _project_early_drop_python(subset, mask, inside, corr, partner, term, accept_prob, n_model):
    """the loop of include/impl/scene.hpp:421-509 with early_out = true, statement by statement, over per-point outcomes
    computed beforehand: the test sits at the END of the body, so masked points and points outside the grid (`continue`,
    :434,:458) never run it; the bound is the literal expression with its negative operands, the casts wrapping as they
    do on x86-64. Returns (scene_corrs, model_corrs, score, saved)."""
    import math
    S = len(subset)
    tests = [int(np.float32(np.float32(np.float32(0.05) * np.float32(i + 1)) * np.float32(S))) for i in range(int(np.float32(1.0) / np.float32(0.05)) - 2)]
    sc, mc = [], []
    tried, next_test, score = 0, 0, 0.0
    thresh = np.float32(accept_prob) * np.float32(n_model)  # float * size_t -> float (:500)
    for idx in subset:
        tried += 1
        if mask[idx]:
            continue
        if not inside[idx]:
            continue
        if corr[idx]:
            score += float(np.float64(term[idx]))
            sc.append(int(idx))
            mc.append(int(partner[idx]))
        if next_test < len(tests) and tried >= tests[next_test]:
            N = -2.0 - tried
            x = -2.0 - S
            n = -1.0 - len(mc)
            tmp = math.sqrt((x * n * (N - x) * (N - n)) / (N - 1.0))
            inner = int((x * n + tmp) / N) % (1 << 32)      # static_cast<uint32_t>(negative double): wraps
            upper = int(-1.0 - inner) % (1 << 32)           # ... and the double result into the uint32_t `upper`
            if np.float32(upper) < thresh:                  # uint32_t < float
                return sc, mc, score, S - tried
            next_test += 1
    return sc, mc, score / n_model, 0


def test_project_with_early_drop_against_a_python_restatement(oracle_small, small_case):
    """The early-drop rule of project_ (include/impl/scene.hpp:421-509) against the loop written a second time in Python
    over numpy's per-point outcomes: correspondences, partial or normalised score and `saved` for neighbourhood subsets of
    several sizes (down to a handful of points, where checkpoints coincide), good, nearly good and wrong poses, several
    accept_prob, with mask_ set."""
    m, res = oracle_small["model"], small_case["resolution"]
    model, sc_ = small_case["model"], small_case["scene"]
    S, tmk = sc_["surfels"], sc_["tangent_mask"]
    s = oracle_small["scene"]
    ext, tv = m.grid()
    vox = m.voxels()
    rng = np.random.default_rng(21)
    mask = (rng.random(len(S)) < 0.2).astype(np.int32)
    poses = []
    for base in sc_["poses_scene_to_model"]:
        poses.append(synth.pose_colmajor(base))
        for shift in (0.5, 2.0, 6.0, 40.0):
            T = np.array(base, np.float64)
            T[:3, 3] += rng.normal(0, shift * res, 3)
            poses.append(synth.pose_colmajor(T))
    firsts = [int(np.nonzero(sc_["instance_id"] == k)[0][5]) for k in (0, 1)]
    subsets = []
    for c in firsts:
        for r in (m.diameter, 0.4 * m.diameter, 6 * res, 0.0):
            subsets.append(s.radius_subset(c, r))
    dropped = full = 0
    try:
        for use_mask in (False, True):
            s.set_mask(mask if use_mask else None)
            mk = mask if use_mask else np.zeros(len(S), np.int32)
            for pose in poses:
                inside, corr, partner, term = _point_outcomes_numpy(model, res, vox, ext, tv, S, tmk, pose, 3.0)
                for sub in subsets:
                    for accept in (0.5, 0.1, 0.9):
                        want = _project_early_drop_python(sub, mk, inside, corr, partner, term, accept, len(model))
                        got = s.project(m, pose, accept, 3.0, subset=sub, early_out=True)
                        assert list(got[0]) == want[0] and list(got[1]) == want[1], (use_mask, len(sub), accept)
                        assert got[3] == want[3] and abs(got[2] - want[2]) <= 1e-12 * max(1.0, abs(want[2]))
                        dropped += int(want[3] > 0)
                        full += int(want[3] == 0 and len(want[0]) > 50)
    finally:
        s.set_mask(None)
    assert dropped > 50 and full > 20
