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
