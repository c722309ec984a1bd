"""GPU parity tests proper: every call goes through the C ABI (libtm_hip.so) and is compared with
the CPU oracle on the same seeded inputs. Bar: bit-exact for keys, hashes, table contents, voxel
indices, inlier counts, correspondences and (here) poses; scores (fp64 sums of fp32 terms,
order-free fixed-point accumulation on the GPU) to 1e-9 relative plus 2^-40 per summed term (raw early-drop sums can be tiny)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import random_poses_near_truth, sample_pairs
from oracle import tmo
from triplet_match_amd import capi, synth

pytestmark = pytest.mark.gpu

DT = 3.0
SCORE_RTOL = 1e-9
SCORE_ATOL_PER_TERM = 1e-12  # each term is truncated to 2^-40 fixed point before the integer sum


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_device_descriptor_kernel_bit_exact(gpu_ctx, small_case):
    model, subset = small_case["model"], small_case["subset"]
    rng = np.random.default_rng(3)
    ab = rng.choice(subset, size=(4000, 2))
    ab = ab[ab[:, 0] != ab[:, 1]]
    bmin = np.array([0.25, 0, 0, 0.25], np.float32)
    bmax = np.array([1.45, 1.6, 1.6, 1.45], np.float32)
    step = capi.default_angle_step()
    feats, keys, hashes = gpu_ctx.features_batch(model, ab, bmin, bmax, 20.0, step)
    for t in range(len(ab)):
        f = tmo.feature(model[ab[t, 0]], model[ab[t, 1]])
        assert np.array_equal(bits(feats[t]), bits(f)), t
        k = tmo.discretize_feature(f, bmin, bmax, 20.0, step)
        assert np.array_equal(keys[t], k), t
        assert hashes[t] == tmo.murmur_key(k)


def test_descriptor_bin_edges(gpu_ctx):
    """perpendicular / parallel configurations sit exactly on angle-bin edges."""
    s = np.zeros((4, 12), np.float32)
    s[1, 0] = 1.0
    s[2, 1] = 0.5
    s[3, 0:3] = [0.3, 0.3, 0.0]
    s[0, 9:12] = [0, 1, 0]
    s[1, 9:12] = [1, 0, 0]
    s[2, 9:12] = [0, 0, 1]
    s[3, 9:12] = np.array([1, 1, 0], np.float32) / np.float32(np.sqrt(2))
    ab = np.array([[a, b] for a in range(4) for b in range(4) if a != b], np.uint32)
    bmin = np.array([0.1, 0, 0, 0.1], np.float32)
    bmax = np.array([1.2, 1.6, 1.6, 1.2], np.float32)
    step = float(np.float32(0.17453294))
    feats, keys, hashes = gpu_ctx.features_batch(s, ab, bmin, bmax, 20.0, step)
    for t in range(len(ab)):
        f = tmo.feature(s[ab[t, 0]], s[ab[t, 1]])
        assert np.array_equal(bits(feats[t]), bits(f))
        assert np.array_equal(keys[t], tmo.discretize_feature(f, bmin, bmax, 20.0, step))
    assert keys[0][1] == 9  # 90 degrees / 10 degrees lands on bin 9 exactly


def test_voxel_grid_bit_exact(gpu_small, oracle_small):
    g, o = gpu_small["model"], oracle_small["model"]
    eg, tg = g.grid()
    eo, to = o.grid()
    assert np.array_equal(eg, eo) and np.array_equal(bits(tg), bits(to))
    assert abs(g.diameter - o.diameter) == 0.0
    vg, vo = g.voxels(), o.voxels()
    assert vg.shape == vo.shape
    assert np.array_equal(vg, vo)
    rng = np.random.default_rng(0)
    for _ in range(50):
        p = np.append(rng.random(3) * 1.4 - 0.2, 1.0).astype(np.float32)
        assert g.voxel_query(p) == o.voxel_query(p)


def test_grid_over_the_callers_subset(gpu_ctx, small_case):
    """init(subset, ...) with a non-empty subset: bounding box, diameter and voxel grid are taken over the
    finite points of THAT subset (include/impl/model.hpp:17-38), the voxel NN still over the whole cloud."""
    model, res = small_case["model"], small_case["resolution"]
    sub = np.nonzero(model[:, 0] < np.median(model[:, 0]))[0].astype(np.uint32)
    g = capi.Model(gpu_ctx, model, res)
    o = tmo.Model(model, res)
    full_diam = g.diameter
    g.set_grid_subset(sub)
    o.set_grid_subset(sub)
    assert g.diameter < full_diam and g.diameter == o.diameter
    g.build_voxels()
    o.build_voxels(threads=8)
    eg, tg = g.grid()
    eo, to = o.grid()
    assert np.array_equal(eg, eo) and np.array_equal(bits(tg), bits(to))
    vg = g.voxels()
    assert np.array_equal(vg, o.voxels())
    assert (~np.isin(vg, sub)).any()  # nearest points outside the subset do appear: whole-cloud NN
    g.set_grid_subset(None)  # back to the default
    assert g.diameter == full_diam


def test_voxel_grid_with_ties_and_degenerate_axis(gpu_ctx):
    """planar cloud (one axis has range 0 -> scale 1, ext 1) with duplicated points (ties)."""
    rng = np.random.default_rng(5)
    n = 400
    s = np.zeros((n, 12), np.float32)
    s[:, 0:2] = rng.random((n, 2))
    s[:, 3] = 1
    s[:, 6] = 1
    s[n // 2:, 0:3] = s[:n // 2, 0:3]  # exact duplicates: lowest index must win
    res = 0.05
    g = capi.Model(gpu_ctx, s, res)
    g.build_voxels()
    o = tmo.Model(s, res)
    o.build_voxels(threads=4, brute=True)
    assert np.array_equal(g.grid()[0], o.grid()[0])
    vg, vo = g.voxels(), o.voxels()
    assert np.array_equal(vg, vo) and vg.max() < n // 2
    # the model's arrays, order and tree built on the host (round 1-3) instead of on the device: same table; the same
    # with a few non-finite points in the cloud and with clouds of one to nine points (a single leaf, no tree at all)
    s2 = s.copy()
    s2[[3, 77, 399], 0] = np.nan
    s2[5, 2] = np.inf
    for cloud in (s, s2, s[:1], s[:2], s[:9], s2[:6]):
        ref = tmo.Model(cloud, res)
        ref.build_voxels(threads=4, brute=True)
        for host_build in (0, 1):
            with gpu_ctx.option("model_host_build", host_build):
                gm = capi.Model(gpu_ctx, cloud, res)
                gm.build_voxels()
                assert np.array_equal(gm.voxels(), ref.voxels()), (len(cloud), host_build)
                gm.close()


def test_pair_table_bit_exact(gpu_small, oracle_small, small_case):
    g, o = gpu_small["model"], oracle_small["model"]
    bg, bo = g.feature_bounds(), o.feature_bounds()
    assert np.array_equal(bits(bg[0]), bits(bo[0])) and np.array_equal(bits(bg[1]), bits(bo[1]))
    pc = g.pair_counts()
    assert pc["entries"] == o.pair_count() and pc["valid_pairs"] == o.valid_count()
    keys, offs, pairs = g.table()
    okeys, ocounts = o.table_keys()
    assert np.array_equal(keys, okeys)
    assert np.array_equal(np.diff(offs.astype(np.int64)), ocounts.astype(np.int64))
    for k in range(len(keys)):
        ent, total = o.query_key(keys[k])
        assert np.array_equal(pairs[offs[k]:offs[k + 1]], ent), keys[k]
    assert g.info()["point_count"] == len(small_case["subset"])


def test_query_matches_container_order_and_limit(gpu_small, oracle_small, small_case):
    g, o = gpu_small["model"], oracle_small["model"]
    model, subset = small_case["model"], small_case["subset"]
    rng = np.random.default_rng(8)
    hits = 0
    for _ in range(60):
        i, j = rng.choice(subset, size=2, replace=False)
        f = tmo.feature(model[i], model[j])
        eo, to = o.query(f, cap=200)
        eg, tg = g.query(f, cap=200)
        assert tg == to and np.array_equal(eg, eo)
        hits += to > 0
    assert hits > 10
    # a key that cannot exist
    eg, tg = g.query_key([19, 9, 9, 18])
    assert tg == 0 and len(eg) == 0


def test_query_uninitialised_model_errors(gpu_ctx, small_case):
    m = capi.Model(gpu_ctx, small_case["model"], small_case["resolution"])
    with pytest.raises(capi.TmError) as e:
        m.query([1.0, 0.5, 0.5, 1.0])
    assert e.value.code == capi.ERR_NOT_INITIALIZED and "uninitialized" in str(e.value)
    with pytest.raises(capi.TmError):
        m.build_pairs(small_case["subset"][:1])  # fewer than two points -> empty table


def test_hypothesis_generation_bit_exact(gpu_small, oracle_small, small_case):
    pairs = sample_pairs(small_case, oracle_small["model"], n_first=40, per_first=8)
    # add pairs that must be filtered: i == j, non-tangent j, out-of-window
    tm = small_case["scene"]["tangent_mask"]
    nont = int(np.nonzero(tm == 0)[0][0])
    extra = np.array([[pairs[0, 0], pairs[0, 0]], [pairs[0, 0], nont]], np.uint32)
    pairs = np.concatenate([extra, pairs])
    po, vo = oracle_small["scene"].generate_hypotheses(oracle_small["model"], pairs)
    pg, vg, total = gpu_small["scene"].generate_hypotheses(gpu_small["model"], pairs)
    assert total == len(po) and len(pg) == len(po) and len(po) > 500
    assert np.array_equal(vg, vo)
    assert np.array_equal(bits(pg), bits(po))  # poses bit for bit
    # capacity truncation keeps the first `cap` in visit order
    pg2, vg2, total2 = gpu_small["scene"].generate_hypotheses(gpu_small["model"], pairs, cap=100)
    assert total2 == total and np.array_equal(vg2, vo[:100])
    # query_limit
    po3, vo3 = oracle_small["scene"].generate_hypotheses(oracle_small["model"], pairs, query_limit=7)
    pg3, vg3, _ = gpu_small["scene"].generate_hypotheses(gpu_small["model"], pairs, query_limit=7)
    assert np.array_equal(vg3, vo3) and vo3[:, 3].max() == 6
    # force_up (ordered compaction)
    po4, vo4 = oracle_small["scene"].generate_hypotheses(oracle_small["model"], pairs, force_up=True)
    pg4, vg4, total4 = gpu_small["scene"].generate_hypotheses(gpu_small["model"], pairs, force_up=True)
    assert total4 == len(po4) < len(po)
    assert np.array_equal(vg4, vo4) and np.array_equal(bits(pg4), bits(po4))


def test_inlier_counts_bit_exact_full_scene(gpu_small, oracle_small, small_case):
    pairs = sample_pairs(small_case, oracle_small["model"], n_first=30, per_first=6)
    poses, _ = oracle_small["scene"].generate_hypotheses(oracle_small["model"], pairs)
    poses = np.concatenate([poses[:1500], random_poses_near_truth(small_case, 40)])
    co, so, _ = oracle_small["scene"].score_hypotheses(oracle_small["model"], poses, 0.5, DT, threads=8)
    cg, sg = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT,
                                                  mode=capi.SCORE_BRUTEFORCE)
    assert np.array_equal(cg, co)
    assert co.max() > 1500  # the planted poses are in the batch
    assert np.allclose(sg, so, rtol=SCORE_RTOL, atol=1e-15)
    cg2, sg2 = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT, mode=capi.SCORE_AUTO)
    assert np.array_equal(cg2, co) and np.allclose(sg2, so, rtol=SCORE_RTOL, atol=1e-15)


def test_inlier_counts_subset_and_mask(gpu_small, oracle_small, small_case):
    poses = random_poses_near_truth(small_case, 30)
    inst0 = np.nonzero(small_case["scene"]["instance_id"] == 0)[0]
    sub = oracle_small["scene"].radius_subset(int(inst0[0]), oracle_small["model"].diameter)
    co, so, _ = oracle_small["scene"].score_hypotheses(oracle_small["model"], poses, 0.5, DT, subset=sub)
    cg, sg = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT, subset=sub)
    assert np.array_equal(cg, co) and np.allclose(sg, so, rtol=SCORE_RTOL, atol=1e-15)
    # the list form (one thread per (pose, point)): same counts and the same score bits
    cp, sp = gpu_small["scene"].score_hypotheses_on_points(gpu_small["model"], poses, sub, DT)
    assert np.array_equal(cp, cg) and np.array_equal(sp, sg)
    cp0, sp0 = gpu_small["scene"].score_hypotheses_on_points(gpu_small["model"], poses, np.zeros(0, np.int32), DT)
    assert not cp0.any() and not sp0.any()
    with pytest.raises(capi.TmError):  # a point listed twice
        gpu_small["scene"].score_hypotheses_on_points(gpu_small["model"], poses, np.array([3, 5, 3], np.int32), DT)
    with pytest.raises(capi.TmError):
        gpu_small["scene"].score_hypotheses_on_points(gpu_small["model"], poses, np.array([-1], np.int32), DT)
    # empty subset
    cg0, sg0 = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT,
                                                    subset=np.zeros(0, np.int32))
    assert not cg0.any() and not sg0.any()
    # mask half of instance 0
    mask = np.zeros(small_case["scene"]["surfels"].shape[0], np.int32)
    mask[inst0[::2]] = 1
    oracle_small["scene"].set_mask(mask)
    gpu_small["scene"].set_mask(mask)
    try:
        co, so, _ = oracle_small["scene"].score_hypotheses(oracle_small["model"], poses, 0.5, DT)
        cg, sg = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT)
        assert np.array_equal(cg, co) and np.allclose(sg, so, rtol=SCORE_RTOL, atol=1e-15)
        # subset + mask_: general call == list form == oracle
        cs, ss = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT, subset=sub)
        cp, sp = gpu_small["scene"].score_hypotheses_on_points(gpu_small["model"], poses, sub, DT)
        cso, _, _ = oracle_small["scene"].score_hypotheses(oracle_small["model"], poses, 0.5, DT, subset=sub)
        assert np.array_equal(cp, cs) and np.array_equal(sp, ss) and np.array_equal(cs, cso)
        # mask_add: the listed points set on top of the current mask == set_mask with the updated mask (counts, scores,
        # radius neighbourhoods, the host-side sample filter), also in two instalments and with repeated indices
        more = inst0[1::4].astype(np.uint32)
        gpu_small["scene"].mask_add(more[: len(more) // 2])
        gpu_small["scene"].mask_add(np.r_[more[len(more) // 2:], more[:3]])
        mask2 = mask.copy()
        mask2[more] = 1
        # the live count (finite, unmasked points) stays on the device after a mask change until it is asked for
        finite = np.isfinite(small_case["scene"]["surfels"][:, 0:3]).all(axis=1)
        assert gpu_small["scene"].live_count() == int(np.count_nonzero(finite & (mask2 == 0)))
        assert gpu_small["scene"].live_count() == int(np.count_nonzero(finite & (mask2 == 0)))  # asked twice: kept
        oracle_small["scene"].set_mask(mask2)
        co2, so2, _ = oracle_small["scene"].score_hypotheses(oracle_small["model"], poses, 0.5, DT)
        ca, sa = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT)
        assert np.array_equal(ca, co2) and np.allclose(sa, so2, rtol=SCORE_RTOL, atol=1e-15)
        gpu_small["scene"].set_mask(mask2)
        cb, sb = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT)
        assert np.array_equal(ca, cb) and np.array_equal(sa, sb)
        gpu_small["scene"].set_mask(None)
        assert gpu_small["scene"].live_count() == int(np.count_nonzero(finite))
        gpu_small["scene"].set_mask(None)  # clearing a clear mask is a no-op
        gpu_small["scene"].mask_add(np.flatnonzero(mask2).astype(np.uint32))
        assert gpu_small["scene"].live_count() == int(np.count_nonzero(finite & (mask2 == 0)))
        cc, sc_ = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT)
        assert np.array_equal(cc, cb) and np.array_equal(sc_, sb)
    finally:
        oracle_small["scene"].set_mask(None)
        gpu_small["scene"].set_mask(None)
    co, _, _ = oracle_small["scene"].score_hypotheses(oracle_small["model"], poses, 0.5, DT)
    cg, _ = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT)
    assert np.array_equal(cg, co)


def test_dist_thres_boundary(gpu_small, oracle_small, small_case):
    """d2 <= d2max on the GPU must equal !(sqrtf(d2) > thres) on the CPU for any threshold."""
    poses = random_poses_near_truth(small_case, 12)
    for dt in (0.25, 0.5, 1.0, 1.7, 2.3, 6.0):
        co, _, _ = oracle_small["scene"].score_hypotheses(oracle_small["model"], poses, 0.5, dt)
        cg, _ = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, dt)
        assert np.array_equal(cg, co), dt


def test_extract_correspondences_equals_finish_find(gpu_small, oracle_small, small_case):
    for T in small_case["scene"]["poses_scene_to_model"]:
        P = synth.pose_colmajor(T)
        sco, mco, so, _ = oracle_small["scene"].project(oracle_small["model"], P, 0.5, DT)
        scg, mcg, sg = gpu_small["scene"].extract_correspondences(gpu_small["model"], P, DT)
        assert np.array_equal(scg, sco) and np.array_equal(mcg, mco)
        assert abs(sg - so) <= SCORE_RTOL * abs(so)


def test_early_drop_rule_bit_exact(gpu_small, oracle_small, small_case):
    pairs = sample_pairs(small_case, oracle_small["model"], n_first=20, per_first=5)
    poses, _ = oracle_small["scene"].generate_hypotheses(oracle_small["model"], pairs)
    poses = np.concatenate([poses[:300], random_poses_near_truth(small_case, 30)])
    inst0 = np.nonzero(small_case["scene"]["instance_id"] == 0)[0]
    sub = oracle_small["scene"].radius_subset(int(inst0[0]), oracle_small["model"].diameter)
    rng = np.random.default_rng(1)
    sub_shuffled = sub.copy()
    rng.shuffle(sub_shuffled)
    sub_dup = np.concatenate([sub_shuffled[:200], sub_shuffled[:50], sub_shuffled[200:]])  # points named twice
    n_dropped = n_kept = 0
    for subset in (None, sub, sub_shuffled, sub_dup):
        cg, sg, dg, vg = gpu_small["scene"].score_hypotheses_early_drop(
            gpu_small["model"], poses, DT, 0.5, subset=subset)
        if subset is sub_shuffled:  # the tile-culled bit kernel and the one-thread-per-position kernel agree
            with gpu_small["scene"].ctx.option("ed_bruteforce", 1):
                ref = gpu_small["scene"].score_hypotheses_early_drop(gpu_small["model"], poses, DT, 0.5, subset=subset)
            assert all(np.array_equal(x, y) for x, y in zip((cg, sg, dg, vg), ref))
        for h in range(len(poses)):
            sc, mc, score, saved = oracle_small["scene"].project(
                oracle_small["model"], poses[h], 0.5, DT, subset=subset, early_out=True)
            assert cg[h] == len(sc), (h, cg[h], len(sc))
            assert vg[h] == saved
            if saved > 0:
                assert dg[h] >= 0
            assert abs(sg[h] - score) <= SCORE_RTOL * abs(score) + SCORE_ATOL_PER_TERM * max(1, len(sc))
            n_dropped += dg[h] >= 0
            n_kept += dg[h] < 0
    assert n_dropped > 50 and n_kept > 50  # both outcomes are exercised


def test_best_of_batch(gpu_ctx, gpu_small, small_case):
    poses = random_poses_near_truth(small_case, 64)
    n = len(poses)
    d_p = gpu_ctx.alloc(n * 64)
    d_c = gpu_ctx.alloc(n * 4)
    d_s = gpu_ctx.alloc(n * 8)
    try:
        gpu_ctx.h2d(d_p, poses)
        gpu_small["scene"].score_hypotheses_dev(gpu_small["model"], d_p, n, DT, d_c, d_s)
        counts = np.zeros(n, np.uint32)
        scores = np.zeros(n, np.float64)
        gpu_ctx.d2h(counts, d_c)
        gpu_ctx.d2h(scores, d_s)
        rec = gpu_ctx.best_of_batch(d_c, d_s, d_p, n, id_offset=1000, id_stride=3)
        order = sorted(range(n), key=lambda i: (-int(counts[i]), -scores[i], i))
        assert rec.hyp_id == 1000 + 3 * order[0] and rec.inlier_count == counts[order[0]]
        assert np.array_equal(np.array(rec.pose[:], np.float32), poses[order[0]])
        cg, sg = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT)
        assert np.array_equal(cg, counts) and np.array_equal(sg, scores)  # deterministic scores
    finally:
        for p in (d_p, d_c, d_s):
            gpu_ctx.free(p)


def test_generate_and_score_device_resident(gpu_ctx, gpu_small, oracle_small, small_case):
    pairs = sample_pairs(small_case, oracle_small["model"], n_first=30, per_first=6)
    cap = 4096
    d_ij = gpu_ctx.alloc(pairs.nbytes)
    d_p = gpu_ctx.alloc(cap * 64)
    d_v = gpu_ctx.alloc(cap * 16)
    d_c = gpu_ctx.alloc(cap * 4)
    d_s = gpu_ctx.alloc(cap * 8)
    try:
        gpu_ctx.h2d(d_ij, pairs)
        n, total = gpu_small["scene"].generate_hypotheses_dev(gpu_small["model"], d_ij, len(pairs),
                                                             d_p, d_v, cap)
        gpu_small["scene"].score_hypotheses_dev(gpu_small["model"], d_p, n, DT, d_c, d_s)
        counts = np.zeros(n, np.uint32)
        gpu_ctx.d2h(counts, d_c)
        po, _ = oracle_small["scene"].generate_hypotheses(oracle_small["model"], pairs)
        co, _, _ = oracle_small["scene"].score_hypotheses(oracle_small["model"], po[:n], 0.5, DT, threads=8)
        assert total == len(po) and np.array_equal(counts, co)
    finally:
        for p in (d_ij, d_p, d_v, d_c, d_s):
            gpu_ctx.free(p)


def test_nonfinite_scene_points_are_ignored(gpu_ctx, gpu_small, oracle_small, small_case):
    s = small_case["scene"]["surfels"].copy()
    s[::97, 0] = np.nan
    s[5::101, 2] = np.inf
    tmask = small_case["scene"]["tangent_mask"]
    gs = capi.Scene(gpu_ctx, s, tmask)
    os_ = tmo.Scene(s, tmask)
    poses = random_poses_near_truth(small_case, 10)
    co, _, _ = os_.score_hypotheses(oracle_small["model"], poses, 0.5, DT)
    cg, _ = gs.score_hypotheses(gpu_small["model"], poses, DT)
    assert np.array_equal(cg, co)


def test_traits_project_cloud_bit_exact(gpu_ctx):
    rng = np.random.default_rng(6)
    n = 3000
    theta = rng.random(n) * 2 * np.pi
    h = rng.random(n) * 2
    r = 0.5 + rng.normal(size=n) * 0.01
    nrm = np.stack([np.cos(theta), np.sin(theta), np.zeros(n)], 1)
    s = np.zeros((n, 12), np.float32)
    s[:, 0:3] = np.array([1, 2, 3]) + r[:, None] * nrm + np.stack([0 * h, 0 * h, h], 1)
    s[:, 3] = 1
    s[:, 4:7] = -nrm
    s[::3, 9:12] = [0, 0, 1]
    ms = np.zeros(capi.STATE_FLOATS, np.float32)
    ms[1] = 0.02
    st = capi.traits_init_from_samples(capi.CYLINDER, ms, np.concatenate([s[0], s[7]]))
    for kind in (capi.IDENTITY, capi.PLANE, capi.PLANE2, capi.CYLINDER):
        out, valid = gpu_ctx.traits_project_cloud(kind, st, s)
        for i in range(0, n, 7):
            uvw = tmo.traits_project(kind, st, s[i, 0:3])
            assert (uvw is not None) == bool(valid[i])
            if uvw is None:
                continue
            assert np.array_equal(bits(out[i, 0:3]), bits(uvw)), (kind, i)
            if np.linalg.norm(s[i, 9:12]) > 0:
                assert np.array_equal(bits(out[i, 9:12]), bits(tmo.traits_tangent(kind, st, s[i])))
                assert np.array_equal(bits(out[i, 4:7]), bits(tmo.traits_normal(kind, st, s[i])))
        assert 0 < valid.sum() <= n


def _cylinder_case(rng, n_model=1500, n_scene=6000):
    """pipe section (radius 0.5, height 2) with seam/flange tangents; scene = rigid copy + clutter"""
    def pipe(n, r=0.5):
        theta = rng.random(n) * 2 * np.pi
        h = rng.random(n) * 2.0
        nrm = np.stack([np.cos(theta), np.sin(theta), np.zeros(n)], 1)
        s = np.zeros((n, 12), np.float32)
        s[:, 0:3] = r * nrm + np.stack([0 * h, 0 * h, h], 1)
        s[:, 3] = 1
        s[:, 4:7] = -nrm
        seam = (np.abs(theta - 1.0) < 0.08) | (np.abs(theta - 4.0) < 0.08)      # two axial seams
        flange = (np.abs(h - 0.6) < 0.03) | (np.abs(h - 1.5) < 0.03)             # two rings
        s[seam, 9:12] = [0, 0, 1]
        tang = np.stack([-np.sin(theta), np.cos(theta), np.zeros(n)], 1)
        s[flange & ~seam, 9:12] = tang[flange & ~seam]
        return s
    model = pipe(n_model)
    inst = pipe(n_model)
    R = synth.random_rotation(rng)
    t = np.array([3.0, -2.0, 1.0])
    inst[:, 0:3] = inst[:, 0:3] @ R.T + t
    inst[:, 4:7] = inst[:, 4:7] @ R.T
    inst[:, 9:12] = inst[:, 9:12] @ R.T
    clutter = np.zeros((n_scene - n_model, 12), np.float32)
    clutter[:, 0:3] = rng.random((n_scene - n_model, 3)) * 6 - np.array([0, 5, 2])
    clutter[:, 3] = 1
    clutter[:, 6] = 1
    scene = np.concatenate([inst, clutter]).astype(np.float32)
    return model, scene, R, t


def _project_cloud_oracle(kind, state, surfels):
    out = np.zeros_like(surfels)
    valid = np.zeros(len(surfels), np.int32)
    for i, s in enumerate(surfels):
        uvw = tmo.traits_project(kind, state, s[0:3])
        if uvw is None:
            continue
        valid[i] = 1
        out[i, 0:3] = uvw
        out[i, 3] = 1
        if np.linalg.norm(s[9:12]) > 0:
            out[i, 9:12] = tmo.traits_tangent(kind, state, s)
            out[i, 4:7] = tmo.traits_normal(kind, state, s)
        else:
            st = np.asarray(state)
            g = st[5:21].reshape(4, 4).T[:3, :3]
            v = g.astype(np.float32) @ s[4:7]
            out[i, 4:7] = v
    return out, valid


@pytest.mark.parametrize("kind", ["cylinder", "plane2"])
def test_probe_projected_equals_the_built_scene(gpu_ctx, kind):
    """tm_scene_probe_projected's two counts == what tm_scene_create_projected + tm_scene_download give: live
    points (tm_scene_live_count) and tangent points within the band that are not masked, with and without mask_."""
    rng = np.random.default_rng(31)
    K = capi.CYLINDER if kind == "cylinder" else capi.PLANE2
    ms = np.zeros(capi.STATE_FLOATS, np.float32)
    ms[1] = 0.05
    if kind == "cylinder":
        _, scene, _, _ = _cylinder_case(rng)
        state = capi.traits_init_from_samples(K, ms, np.concatenate([scene[0], scene[11]]))
        band = 0.05 / float(state[0])
    else:
        scene = np.zeros((6000, 12), np.float32)
        scene[:, 0:3] = rng.random((6000, 3)) * [3.0, 2.0, 0.2]
        scene[:, 3] = 1
        scene[:, 6] = 1
        scene[::7, 9:12] = [1, 0, 0]
        scene[5, 0] = np.nan
        state = capi.traits_init_from_samples(K, ms, np.concatenate([scene[0], scene[700], scene[1400]]))
        band = 0.05
    assert state is not None
    tmask_in = (np.linalg.norm(scene[:, 9:12], axis=1) > 0.7).astype(np.int32)
    gs = capi.Scene(gpu_ctx, scene, tmask_in)
    mask = np.zeros(len(scene), np.int32)
    mask[rng.integers(0, len(scene), len(scene) // 5)] = 1
    for m in (None, mask):
        gs.set_mask(m)
        proj = gs.project(K, state)
        uvw, tm = proj.download()
        mm = np.zeros(len(scene), bool) if m is None else m.astype(bool)
        for b in (band, -1.0, 0.0):
            live, tan = gs.probe_projected(K, state, b)
            assert live == proj.live_count()
            on = np.ones(len(scene), bool) if b < 0 else (np.abs(uvw[:, 2]) <= np.float32(b))
            assert tan == int(((tm != 0) & ~mm & on).sum()), (kind, m is None, b)
        assert live > 0
    gs.set_mask(None)



@pytest.mark.parametrize("kind", ["cylinder", "plane2"])
def test_projected_traits_pipeline(gpu_ctx, kind):
    """configs 3/4 wiring: project model and scene into the projector's uvw space, then the same
    kernels run there with that traits' intrinsic_distance (cylinder: wrap-around, plane2: 2-D)."""
    rng = np.random.default_rng(12)
    K = capi.CYLINDER if kind == "cylinder" else capi.PLANE2
    ms = np.zeros(capi.STATE_FLOATS, np.float32)
    if kind == "cylinder":
        model, scene, R, t = _cylinder_case(rng)
        ms[1] = 0.05
        # model projector: known axis (MSAC fit is out of scope); scene projector: from two samples
        mstate = capi.traits_init_from_samples(K, ms, np.concatenate([model[0], model[11]]))
        sstate = capi.traits_init_from_samples(K, ms, np.concatenate([scene[0], scene[11]]))
        radius = float(mstate[0])
    else:
        n = 1500
        model = np.zeros((n, 12), np.float32)
        model[:, 0:2] = rng.random((n, 2)) * [2.0, 1.0]
        model[:, 2] = rng.normal(size=n) * 0.002
        model[:, 3] = 1
        model[:, 6] = 1
        edge = (model[:, 0] < 0.04) | (model[:, 1] < 0.03) | (np.abs(model[:, 0] - 1.2) < 0.02)
        model[edge & (model[:, 1] >= 0.03), 9:12] = [0, 1, 0]
        model[model[:, 1] < 0.03, 9:12] = [1, 0, 0]
        inst = model.copy()
        inst[:, 0:2] = rng.random((n, 2)) * [2.0, 1.0]
        inst[:, 2] = rng.normal(size=n) * 0.002
        inst[:, 9:12] = 0
        e2 = (inst[:, 0] < 0.04) | (np.abs(inst[:, 0] - 1.2) < 0.02)
        inst[e2 & (inst[:, 1] >= 0.03), 9:12] = [0, 1, 0]
        inst[inst[:, 1] < 0.03, 9:12] = [1, 0, 0]
        R = synth.random_rotation(rng)
        t = np.array([1.0, 2.0, -1.0])
        inst[:, 0:3] = inst[:, 0:3] @ R.T + t
        inst[:, 4:7] = inst[:, 4:7] @ R.T
        inst[:, 9:12] = inst[:, 9:12] @ R.T
        clutter = np.zeros((3000, 12), np.float32)
        clutter[:, 0:3] = rng.random((3000, 3)) * 5
        clutter[:, 3] = 1
        clutter[:, 4] = 1
        scene = np.concatenate([inst, clutter]).astype(np.float32)
        mstate = capi.traits_init_from_samples(K, ms, np.concatenate([model[0], model[700], model[1400]]))
        sstate = capi.traits_init_from_samples(K, ms, np.concatenate([scene[0], scene[700], scene[1400]]))
        radius = 0.0
    assert mstate is not None and sstate is not None
    # GPU projection == oracle projection (bit-exact), then build both sides from the GPU output
    mu, mv = gpu_ctx.traits_project_cloud(K, mstate, model)
    su, sv = gpu_ctx.traits_project_cloud(K, sstate, scene)
    mo, mvo = _project_cloud_oracle(K, mstate, model)
    assert np.array_equal(mv, mvo) and np.array_equal(bits(mu[:, 0:3]), bits(mo[:, 0:3]))
    assert mv.sum() > 0.9 * len(model) and sv.sum() >= 0.2 * len(scene)
    mu, su = mu[mv == 1], su.copy()
    su[sv == 0, 0] = np.nan  # projected-away scene points never take part
    res = synth.resolution(mu)
    subset = np.nonzero(np.linalg.norm(mu[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
    tmask = ((np.linalg.norm(su[:, 9:12], axis=1) > 0.7) & (sv == 1)).astype(np.int32)
    gm = capi.Model(gpu_ctx, mu, res)
    gm.set_metric(K, radius)
    gm.build_voxels()
    gm.build_pairs(subset)
    gs = capi.Scene(gpu_ctx, su, tmask)
    om = tmo.Model(mu, res)
    om.set_metric(K, radius)
    om.build_voxels(threads=4)
    assert om.build_pairs(subset) == 0
    os_ = tmo.Scene(su, tmask)
    assert np.array_equal(gm.voxels(), om.voxels())
    tidx = np.nonzero(tmask)[0]
    pairs = rng.choice(tidx, size=(300, 2)).astype(np.uint32)
    pg, vg, total = gs.generate_hypotheses(gm, pairs)
    po, vo = os_.generate_hypotheses(om, pairs)
    assert total == len(po) and np.array_equal(vg, vo) and np.array_equal(bits(pg), bits(po))
    poses = np.concatenate([po[:600], np.eye(4, dtype=np.float32).reshape(1, 16)])
    co, so, _ = os_.score_hypotheses(om, poses, 0.5, DT, threads=4)
    cg, sg = gs.score_hypotheses(gm, poses, DT, traits=K)
    assert np.array_equal(cg, co) and np.allclose(sg, so, rtol=SCORE_RTOL, atol=1e-15)
    assert co.max() > 0
    # the brick filters under the model's metric (near_metric: 0 never, 1 auto, 2 always), grouped and per-hypothesis
    # kernel, ball-restricted and exhaustive: one answer. A larger threshold moves the codes' cut.
    centers = rng.choice(tidx, size=len(poses)).astype(np.uint32)
    for dt in (DT, 1.0, 9.0):
        cb, sb = gs.score_hypotheses(gm, poses, dt, traits=K, mode=capi.SCORE_BRUTEFORCE)
        cball = None
        for nm in (0, 1, 2):
            with gpu_ctx.option("near_metric", nm):
                c1, s1 = gs.score_hypotheses(gm, poses, dt, traits=K, mode=capi.SCORE_CULLED)
                with gpu_ctx.option("score_grouped", 0):
                    c2, s2 = gs.score_hypotheses(gm, poses, dt, traits=K, mode=capi.SCORE_CULLED)
                c3, s3 = gs.score_hypotheses_ball(gm, poses, centers, gm.diameter, dt)
            assert np.array_equal(c1, cb) and np.array_equal(s1, sb), (kind, dt, nm)
            assert np.array_equal(c2, cb) and np.array_equal(s2, sb), (kind, dt, nm)
            if cball is None:
                cball = (c3, s3)
            assert np.array_equal(c3, cball[0]) and np.array_equal(s3, cball[1]), (kind, dt, nm)
    with pytest.raises(capi.TmError):  # the metric is part of the model: a mismatching kind is refused
        gs.score_hypotheses(gm, poses, DT, traits=capi.IDENTITY)
    if kind == "cylinder":
        # wrap-around: a query at u ~ 0 matches a model point at u ~ 2*pi*r only under the cylinder metric
        gm2 = capi.Model(gpu_ctx, mu, res)
        gm2.build_voxels()
        c_l2, _ = gs.score_hypotheses(gm2, poses, DT)
        assert (cg >= c_l2).all()


def test_opencl_kernel_equivalents(gpu_ctx):
    """HIP twins of the reference's orphaned OpenCL kernels: icp_projection (opencl/icp.cl:1-53 +
    cylinder.cl uv_project) bit-exact against the oracle's transcription, icp_correlation (:55-86)
    summed to the 3x3 cross-covariance against a float64 numpy evaluation."""
    from oracle import tmo as o
    rng = np.random.default_rng(9)
    n, W, H = 5000, 96, 64
    theta = rng.random(n) * 2 * np.pi
    pts = np.stack([np.cos(theta) * (1 + rng.normal(size=n) * 0.01), np.sin(theta) * (1 + rng.normal(size=n) * 0.01),
                    rng.random(n), np.ones(n)], 1).astype(np.float32)
    image = np.zeros((W * H, 4), np.float32)
    image[:, 0] = np.tile((np.arange(W) + 0.5) / W, H)
    image[:, 1] = np.repeat((np.arange(H) + 0.5) / H, W)
    eye = np.eye(4, dtype=np.float32).T.reshape(16)
    al = np.eye(4, dtype=np.float32)
    al[:3, :3] = synth.random_rotation(rng, upright=True)
    al = np.ascontiguousarray(al.T.reshape(16))
    size = np.array([W, H], np.int32)
    margin = np.array([2, 3], np.int32)
    out, mi, si, cnt = capi.cl_icp_projection(gpu_ctx, pts, image, size, margin, al, eye, eye, eye, 0.02)
    exp_out = np.zeros_like(out)
    exp_mi = np.zeros(n, np.int32)
    exp_si = np.zeros(n, np.int32)
    L = o.lib()
    import ctypes as C
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    L.tmo_cl_icp_projection(fp(pts), n, fp(image), ip(size), ip(margin), fp(al), fp(eye), fp(eye), fp(eye),
                            C.c_float(0.02), fp(exp_out), ip(exp_mi), ip(exp_si))
    assert np.array_equal(mi, exp_mi) and np.array_equal(si, exp_si)
    assert np.array_equal(bits(out), bits(exp_out))
    assert cnt == int((exp_mi >= 0).sum()) and 0 < cnt < n
    # correlation
    scene4 = rng.normal(size=(4000, 4)).astype(np.float32)
    model4 = rng.normal(size=(3000, 4)).astype(np.float32)
    k = 2500
    ia = rng.integers(0, 4000, k).astype(np.int32)
    ib = rng.integers(0, 3000, k).astype(np.int32)
    cs = scene4[ia].mean(0).astype(np.float32)
    cm = model4[ib].mean(0).astype(np.float32)
    cov = capi.cl_icp_correlation(gpu_ctx, scene4, model4, ia, ib, cs, cm)
    S = (scene4[ia, :3].astype(np.float32) - cs[:3]).astype(np.float64)
    M = (model4[ib, :3].astype(np.float32) - cm[:3]).astype(np.float64)
    ref = (M.T @ S) / (k - 1)  # cov9[3*r + c] = model r x scene c
    assert np.allclose(cov.reshape(3, 3), ref, rtol=1e-5, atol=1e-7)  # fp32 products, tolerance 1e-5


def test_voxel_kdtree_equals_bruteforce_kernel(gpu_ctx):
    """the default kd-tree voxel build and the LDS-tiled exhaustive kernel give the same table
    (20k-point model, ~5e6 voxels), including duplicated points (ties -> lowest original index)."""
    import os
    model, _ = synth.synth_model(20000, seed=4)
    model[15000:15500, 0:3] = model[100:600, 0:3]  # exact duplicates at higher indices
    res = synth.resolution(model)
    a = capi.Model(gpu_ctx, model, res)
    a.build_voxels()
    va = a.voxels()
    with gpu_ctx.option("voxel_bruteforce", 1):
        b = capi.Model(gpu_ctx, model, res)
        b.build_voxels()
        vb = b.voxels()
    assert va.size > 4_000_000 and np.array_equal(va, vb)
    with gpu_ctx.option("model_host_build", 1):  # arrays, curve order and a median kd-tree from the host
        h = capi.Model(gpu_ctx, model, res)
        h.build_voxels()
        assert np.array_equal(va, h.voxels())
        sub_t = np.nonzero(np.linalg.norm(model[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
        a.build_pairs(sub_t)
        h.build_pairs(sub_t)
        ka, kh = a.table(), h.table()
        assert all(np.array_equal(x, y) for x, y in zip(ka, kh)) and len(ka[0]) > 10  # same order -> same table
        h.close()
    with gpu_ctx.option("voxel_seed", 0):  # the packets start without the super-brick's candidate
        u = capi.Model(gpu_ctx, model, res)
        u.build_voxels()
        assert np.array_equal(va, u.voxels())
        u.close()
    with gpu_ctx.option("voxel_packet", 0):  # every voxel on its own stack instead of a brick's 64 as one packet
        c = capi.Model(gpu_ctx, model, res)
        c.build_voxels()
        assert np.array_equal(va, c.voxels())
        sc = capi.Scene(gpu_ctx, model[:256], np.zeros(256, np.int32))
        assert np.array_equal(sc.score_hypotheses(c, np.eye(4, dtype=np.float32).reshape(1, 16), DT)[0],
                              sc.score_hypotheses(a, np.eye(4, dtype=np.float32).reshape(1, 16), DT)[0])  # brick minima too
        c.close()
    # the option toggled BETWEEN create and build_voxels: the model was created with the deep device-built tree, which
    # the per-voxel walk's 24-entry stack must never descend (it would drop subtrees without a word): same table
    t = capi.Model(gpu_ctx, model, res)
    with gpu_ctx.option("voxel_packet", 0):
        t.build_voxels()
    assert np.array_equal(va, t.voxels())
    t.close()
    assert not np.isin(va, np.arange(15000, 15500)).any()  # duplicates never win over their originals
    poses = np.eye(4, dtype=np.float32).reshape(1, 16)
    s = capi.Scene(gpu_ctx, model, (np.linalg.norm(model[:, 9:12], axis=1) > 0.7).astype(np.int32))
    ca, _ = s.score_hypotheses(a, poses, DT)
    cb, _ = s.score_hypotheses(b, poses, DT)
    assert np.array_equal(ca, cb) and ca[0] > 19000  # brick codes agree as well


def test_ball_restricted_batch_scoring(gpu_small, oracle_small, small_case):
    """one launch scores every hypothesis on radius_search_inclusive(diameter, p1) of ITS first sample
    point — the set find_in_subset hands to project_ (include/impl/scene.hpp:273,326)."""
    om, os_ = oracle_small["model"], oracle_small["scene"]
    pairs = sample_pairs(small_case, om, n_first=12, per_first=4)
    poses, prov = os_.generate_hypotheses(om, pairs)
    poses, prov = poses[:1500], prov[:1500]
    centers = pairs[prov[:, 0], 0].astype(np.uint32)
    radius = om.diameter
    cg, sg = gpu_small["scene"].score_hypotheses_ball(gpu_small["model"], poses, centers, radius, DT)
    co = np.zeros(len(poses), np.uint32)
    so = np.zeros(len(poses), np.float64)
    for c in np.unique(centers):
        sel = np.nonzero(centers == c)[0]
        sub = os_.radius_subset(int(c), radius)
        cc, ss, _ = os_.score_hypotheses(om, poses[sel], 0.5, DT, subset=sub, threads=4)
        co[sel], so[sel] = cc, ss
    assert np.array_equal(cg, co) and np.allclose(sg, so, rtol=SCORE_RTOL, atol=1e-15)
    full, _ = gpu_small["scene"].score_hypotheses(gpu_small["model"], poses, DT)
    assert (cg <= full).all() and co.max() > 1000  # with r = diameter the ball rarely cuts an in-grid point
    # a small radius really restricts
    c2, _ = gpu_small["scene"].score_hypotheses_ball(gpu_small["model"], poses[:50], centers[:50], 0.1 * radius, DT)
    sub_small = {int(c): os_.radius_subset(int(c), 0.1 * radius) for c in np.unique(centers[:50])}
    for h in range(50):
        cc, _, _ = os_.score_hypotheses(om, poses[h:h + 1], 0.5, DT, subset=sub_small[int(centers[h])])
        assert c2[h] == cc[0]
    assert (c2 < full[:50]).any()


def test_multi_model_one_scene(gpu_ctx):
    """config-5 shape of work: several models against one scene, each with its own best record.
    Two differently stretched models; the scene holds two instances of A and one of B."""
    A, subA = synth.synth_model(1500, seed=1)
    B, subB = synth.synth_model(1500, seed=7, scale=(0.7, 1.3, 1.1))
    resA, resB = synth.resolution(A), synth.resolution(B)
    sc = synth.synth_scene(12000, 1500, seed=2, copies=2, model_resolution=resA)
    rng = np.random.default_rng(3)
    instB, _ = synth.synth_model(1500, seed=8, scale=(0.7, 1.3, 1.1))
    R = synth.random_rotation(rng)
    t = np.array([4.0, 3.0, 1.5])
    instB[:, 0:3] = instB[:, 0:3] @ R.T + t
    instB[:, 4:7] = instB[:, 4:7] @ R.T
    instB[:, 9:12] = instB[:, 9:12] @ R.T
    surf = np.concatenate([sc["surfels"], instB]).astype(np.float32)
    tmask = (np.linalg.norm(surf[:, 9:12], axis=1) > 0.7).astype(np.int32)
    TB = np.eye(4)
    TB[:3, :3] = R
    TB[:3, 3] = t
    truthA = [synth.pose_colmajor(T) for T in sc["poses_scene_to_model"]]
    truthB = [synth.pose_colmajor(np.linalg.inv(TB))]
    poses = np.array(truthA + truthB, np.float32)
    gs = capi.Scene(gpu_ctx, surf, tmask)
    os_ = tmo.Scene(surf, tmask)
    best = {}
    for name, M, sub, res in (("A", A, subA, resA), ("B", B, subB, resB)):
        gm = capi.Model(gpu_ctx, M, res)
        gm.build_voxels()
        gm.build_pairs(sub)
        om = tmo.Model(M, res)
        om.build_voxels(threads=4)
        assert om.build_pairs(sub) == 0
        cg, sg = gs.score_hypotheses(gm, poses, DT)
        co, so, _ = os_.score_hypotheses(om, poses, 0.5, DT)
        assert np.array_equal(cg, co) and np.allclose(sg, so, rtol=SCORE_RTOL, atol=1e-15)
        best[name] = cg
    # each model recognises its own instances only
    assert best["A"][:2].min() > 1100 and best["A"][2] < 0.8 * best["A"][:2].min()
    assert best["B"][2] > 1100 and best["B"][:2].max() < 0.8 * best["B"][2]
    recs = []
    for k, name in enumerate(("A", "B")):
        i = int(np.argmax(best[name]))
        r = capi.BestRecord()
        r.inlier_count, r.hyp_id, r.score = int(best[name][i]), i, 0.0
        recs.append(r)
    assert recs[0].hyp_id in (0, 1) and recs[1].hyp_id == 2  # per-model best records


def test_edge_cases(gpu_ctx, gpu_small, oracle_small, small_case):
    gm, gs = gpu_small["model"], gpu_small["scene"]
    om, os_ = oracle_small["model"], oracle_small["scene"]
    poses = random_poses_near_truth(small_case, 8)
    # empty batches
    c, s = gs.score_hypotheses(gm, np.zeros((0, 16), np.float32), DT)
    assert len(c) == 0 and len(s) == 0
    p, v, total = gs.generate_hypotheses(gm, np.zeros((0, 2), np.uint32))
    assert len(p) == 0 and total == 0
    # out-of-range and degenerate sample pairs are filtered, not faulted
    n = small_case["scene"]["surfels"].shape[0]
    bad = np.array([[n + 5, 0], [0, n + 7], [3, 3]], np.uint32)
    p, v, total = gs.generate_hypotheses(gm, bad)
    assert total == 0
    # thresholds: huge (brick filter inert), zero and negative
    for dt in (100.0, 0.0, -1.0):
        co, _, _ = os_.score_hypotheses(om, poses, 0.5, dt)
        for mode in (capi.SCORE_AUTO, capi.SCORE_BRUTEFORCE):
            cg, _ = gs.score_hypotheses(gm, poses, dt, mode=mode)
            assert np.array_equal(cg, co), (dt, mode)
    # NaN / non-rigid poses: defined behaviour, identical on both paths
    weird = poses.copy()
    weird[0, 5] = np.nan
    weird[1, :] *= 1.5  # scaled: not a rigid motion, still a valid 4x4
    weird[1, 15] = 1.0
    weird[2, 12:15] = 1e6
    co, _, _ = os_.score_hypotheses(om, weird, 0.5, DT)
    for mode in (capi.SCORE_AUTO, capi.SCORE_BRUTEFORCE):
        cg, _ = gs.score_hypotheses(gm, weird, DT, mode=mode)
        assert np.array_equal(cg, co), mode
    # everything masked
    mask = np.ones(n, np.int32)
    gs.set_mask(mask)
    try:
        cg, sg = gs.score_hypotheses(gm, poses, DT)
        assert not cg.any() and not sg.any()
        sc, mc, score = gs.extract_correspondences(gm, poses[0], DT)
        assert len(sc) == 0 and score == 0.0
    finally:
        gs.set_mask(None)
    # early drop on an empty subset
    cnt, sc, dr, sv = gs.score_hypotheses_early_drop(gm, poses, DT, 0.5, subset=np.zeros(0, np.int32))
    assert not cnt.any() and (dr == -1).all() and not sv.any()
    # a two-point tangent subset builds (or reports "no valid pair") without faulting
    tiny = capi.Model(gpu_ctx, small_case["model"], small_case["resolution"])
    try:
        tiny.build_pairs(small_case["subset"][:2])
        assert tiny.pair_counts()["entries"] <= 2
    except capi.TmError as e:
        assert e.code == capi.ERR_EMPTY


def test_cloud_resolution_bit_exact(gpu_ctx, curv_walk):
    """cloud->resolution(): GPU 2-NN distances and the f32 running mean equal the oracle's exhaustive
    version bit for bit (duplicates -> distance 0, non-finite points skipped)."""
    model, _ = synth.synth_model(6000, seed=5)
    model[100:110, 0:3] = model[200:210, 0:3]
    model[50, 0] = np.nan
    ro, do = tmo.cloud_resolution(model)
    for host_trees in (0, 1):  # order + tree built on the device (default) / median kd-tree built on host threads
        with gpu_ctx.option("model_host_build", host_trees):
            rg, dg = gpu_ctx.cloud_resolution(model)
            assert np.array_equal(dg.view(np.uint32), do.view(np.uint32)), host_trees
            assert np.float32(rg).view(np.uint32) == np.float32(ro).view(np.uint32)
            assert (dg[100:110] == 0).all() and abs(rg - synth.resolution(model[np.isfinite(model[:, 0])])) < 1e-4
            for tiny in (model[:2], model[48:52], model[:9]):  # a single leaf
                rt, dt_ = gpu_ctx.cloud_resolution(tiny)
                rto, dto = tmo.cloud_resolution(tiny)
                assert np.float32(rt).view(np.uint32) == np.float32(rto).view(np.uint32), (host_trees, len(tiny))
                assert np.array_equal(dt_.view(np.uint32), dto.view(np.uint32))
            # fewer than two finite points: the reference indexes past its one neighbour (undefined,
            # include/impl/pointcloud.hpp:73-78); here the resolution is 0
            assert gpu_ctx.cloud_resolution(model[:1])[0] == 0.0 and gpu_ctx.cloud_resolution(model[50:52])[0] == 0.0


@pytest.fixture(params=[1, 0], ids=["eight_lanes_per_query", "thread_per_query"])
def curv_walk(request, gpu_ctx):
    """the k-nearest lists of the curvature pass, of cloud->resolution() and of the neighbour accessors: eight lanes per
    query with the list in registers (default) or one thread per query with the list in LDS (curv_groups = 0) — the
    same lists either way"""
    old = gpu_ctx.get_option("curv_groups")
    gpu_ctx.set_option("curv_groups", request.param)
    yield request.param
    gpu_ctx.set_option("curv_groups", old)


def test_cloud_curvature_bit_exact(gpu_ctx, curv_walk):
    """cloud->curvature(30, i): kNN-30 + projected-normal covariance + Jacobi eigenvalues, GPU vs oracle."""
    model, _ = synth.synth_model(3000, seed=6)
    model[10:14, 0:3] = model[20:24, 0:3]  # duplicates: ties in the neighbour order
    model[77, 1] = np.inf
    co = tmo.cloud_curvature(model, 30)
    with gpu_ctx.option("model_host_build", 1):  # the median kd-tree built on host threads (rounds 1-3)
        ch = gpu_ctx.cloud_curvature(model, 30)
        sub = np.array([77, 5, 2999, 10, 11, 12, 13, 20], np.uint32)
        assert np.array_equal(gpu_ctx.cloud_curvature_subset(model, sub, 30).view(np.uint32), co[sub].view(np.uint32))
    assert np.array_equal(ch.view(np.uint32), co.view(np.uint32))
    cg = gpu_ctx.cloud_curvature(model, 30)  # default: curve order and tree made on the device
    assert np.array_equal(cg.view(np.uint32), co.view(np.uint32))
    assert np.array_equal(gpu_ctx.cloud_curvature_subset(model, sub, 30).view(np.uint32), co[sub].view(np.uint32))
    for n_tiny in (1, 2, 9, 40):  # fewer points than k, a single leaf
        assert np.array_equal(gpu_ctx.cloud_curvature(model[:n_tiny], 30).view(np.uint32),
                              tmo.cloud_curvature(model[:n_tiny], 30).view(np.uint32)), n_tiny
    # edge-like points (normals of two faces among the neighbours) have pc_min / pc_max < 0.2
    with np.errstate(invalid="ignore", divide="ignore"):
        ratio = cg[:, 3] / cg[:, 4]
    edge = ratio < 0.2
    assert 50 < edge.sum() < 2500 and cg[77].sum() == 0
    for k in (1, 5, 32):
        assert np.array_equal(gpu_ctx.cloud_curvature(model[:500], k).view(np.uint32),
                              tmo.cloud_curvature(model[:500], k).view(np.uint32))


def test_radius_subset_bit_exact(gpu_ctx, gpu_small, oracle_small, small_case):
    """nn = radius_search_inclusive(diameter, p1) (include/impl/scene.hpp:273): same members, ascending."""
    gs, os_ = gpu_small["scene"], oracle_small["scene"]
    diam = oracle_small["model"].diameter
    rng = np.random.default_rng(21)
    n = len(small_case["scene"]["surfels"])
    for c in rng.integers(0, n, 12):
        for r in (diam, 0.25 * diam, 0.0):
            a, b = gs.radius_subset(int(c), r), os_.radius_subset(int(c), r)
            assert np.array_equal(a, b)
            assert int(c) in a  # inclusive
    # points exactly on the sphere, and non-finite points
    s = np.zeros((300, 12), np.float32)
    s[:, 3] = 1
    s[1:101, 0] = 0.5
    s[101:201, 1] = np.nextafter(np.float32(0.5), np.float32(1))
    s[201:250, 2] = np.nextafter(np.float32(0.5), np.float32(0))
    s[250, 0] = np.nan
    s[251, 1] = np.inf
    gs2 = capi.Scene(gpu_ctx, s, np.zeros(300, np.int32))
    os2 = tmo.Scene(s, np.zeros(300, np.int32))
    a, b = gs2.radius_subset(0, 0.5), os2.radius_subset(0, 0.5)
    assert np.array_equal(a, b) and len(a) == 1 + 100 + 49 + 48  # centre, on-sphere, just inside, origin


def test_tiny_scenes_and_nonfinite_poses(gpu_ctx, gpu_small, oracle_small, small_case):
    """Scenes smaller than a tile / a quarter tile, and poses with NaN / inf entries or a zero rotation."""
    gm, om = gpu_small["model"], oracle_small["model"]
    S, tmk = small_case["scene"]["surfels"], small_case["scene"]["tangent_mask"]
    inst = np.nonzero(small_case["scene"]["instance_id"] == 0)[0]
    poses = random_poses_near_truth(small_case, 6)
    for n in (1, 15, 17, 63, 65, 1025):
        idx = inst[:n]
        gs = capi.Scene(gpu_ctx, S[idx], tmk[idx])
        os_ = tmo.Scene(S[idx], tmk[idx])
        cg, sg = gs.score_hypotheses(gm, poses, DT)
        co, so, _ = os_.score_hypotheses(om, poses, 0.5, DT)
        assert np.array_equal(cg, co), n
        assert np.allclose(sg, so, rtol=SCORE_RTOL, atol=1e-15)
        cb, _ = gs.score_hypotheses(gm, poses, DT, mode=capi.SCORE_BRUTEFORCE)
        assert np.array_equal(cb, co)
        a, b = gs.radius_subset(0, 1e9), os_.radius_subset(0, 1e9)
        assert np.array_equal(a, b) and len(a) == n
    bad = poses[:5].copy()
    bad[0, 12] = np.nan        # NaN translation
    bad[1, 0] = np.inf         # inf in the rotation block
    bad[2, :] = 0.0            # all-zero matrix: every point lands on the origin
    bad[3, 13] = 1e30          # far away
    bad[4, :12] *= 1e-20       # collapsed rotation
    gs, os_ = gpu_small["scene"], oracle_small["scene"]
    cg, sg = gs.score_hypotheses(gm, bad, DT)
    co, so, _ = os_.score_hypotheses(om, bad, 0.5, DT)
    assert np.array_equal(cg, co)
    assert np.allclose(sg, so, rtol=SCORE_RTOL, atol=1e-15, equal_nan=True)
    cb, _ = gs.score_hypotheses(gm, bad, DT, mode=capi.SCORE_BRUTEFORCE)
    assert np.array_equal(cb, co)


def test_early_drop_ball_batch_equals_per_neighbourhood_calls(gpu_small, oracle_small, small_case):
    """One call for hypotheses that each scan their own radius neighbourhood == one early-drop call per
    neighbourhood == the oracle's project_ with early-out on nn(centre)."""
    gs, gm = gpu_small["scene"], gpu_small["model"]
    os_, om = oracle_small["scene"], oracle_small["model"]
    pairs = sample_pairs(small_case, om, n_first=20, per_first=5)
    poses, _ = os_.generate_hypotheses(om, pairs)
    poses = np.concatenate([poses[:240], random_poses_near_truth(small_case, 24)])
    rng = np.random.default_rng(4)
    tidx = np.nonzero(small_case["scene"]["tangent_mask"])[0]
    cset = rng.choice(tidx, size=5, replace=False).astype(np.uint32)
    centers = cset[rng.integers(0, 5, len(poses))]
    diam = om.diameter
    mask = np.zeros(len(small_case["scene"]["surfels"]), np.int32)
    mask[rng.integers(0, len(mask), 800)] = 1
    for use_mask in (False, True):
        gs.set_mask(mask if use_mask else None)
        os_.set_mask(mask if use_mask else None)
        try:
            cb, sb, db, vb = gs.score_hypotheses_early_drop_ball(gm, poses, centers, diam, DT, 0.5)
            for c in cset:
                sel = np.nonzero(centers == c)[0]
                nn = gs.radius_subset(int(c), diam)
                c1, s1, d1, v1 = gs.score_hypotheses_early_drop(gm, poses[sel], DT, 0.5, subset=nn)
                assert np.array_equal(cb[sel], c1) and np.array_equal(sb[sel], s1)
                assert np.array_equal(db[sel], d1) and np.array_equal(vb[sel], v1)
                for h in sel[:12]:
                    sc, mc, score, saved = os_.project(om, poses[h], 0.5, DT, subset=nn, early_out=True)
                    assert cb[h] == len(sc) and vb[h] == saved
                    assert abs(sb[h] - score) <= SCORE_RTOL * abs(score) + SCORE_ATOL_PER_TERM * max(1, len(sc))
            assert (db >= 0).any() and (db < 0).any()
            # the first step from the neighbourhood list in scan order (instead of the same positions in curve order)
            with gs.ctx.option("ed_prefix_list", 0):
                alt = gs.score_hypotheses_early_drop_ball(gm, poses, centers, diam, DT, 0.5)
            assert all(np.array_equal(x, y) for x, y in zip((cb, sb, db, vb), alt))
            # the continuation on bit rows instead of the interval reductions; the first step one workgroup per
            # hypothesis instead of sixteen hypotheses of a neighbourhood per workgroup
            for opts in ({"ed_hist": 0}, {"ed_prefix_group": 0}, {"ed_hist": 0, "ed_prefix_group": 0}, {"ed_ball_list": 0},
                         {"ed_prefix_group": 16}):
                old_opts = {k: gs.ctx.get_option(k) for k in opts}
                for k, v in opts.items():
                    gs.ctx.set_option(k, v)
                try:
                    alt = gs.score_hypotheses_early_drop_ball(gm, poses, centers, diam, DT, 0.5)
                finally:
                    for k, v in old_opts.items():
                        gs.ctx.set_option(k, v)
                assert all(np.array_equal(x, y) for x, y in zip((cb, sb, db, vb), alt)), opts
        finally:
            gs.set_mask(None)
            os_.set_mask(None)


def test_early_drop_interval_reductions_and_their_fallback(gpu_ctx, gpu_small, oracle_small, small_case, capfd):
    """The continuation of the rule on interval reductions (k_ed_hist_wide) == the bit-array kernels == the oracle,
    also where the reductions cannot decide and hand the row over: nothing is ever dropped (accept_prob 0) so every
    checkpoint is visited, poses with a handful of in-grid points leave intervals without a test position, and
    neighbourhoods of fewer than 20 points have coinciding checkpoints."""
    gs, gm = gpu_small["scene"], gpu_small["model"]
    os_, om = oracle_small["scene"], oracle_small["model"]
    rng = np.random.default_rng(77)
    near = random_poses_near_truth(small_case, 40, seed=5)
    far = near.copy()
    far[:, 12:15] += rng.normal(0, 0.45 * om.diameter, (len(far), 3)).astype(np.float32)  # mostly outside the grid
    poses = np.concatenate([near, far])
    tidx = np.nonzero(small_case["scene"]["tangent_mask"])[0]
    cset = rng.choice(tidx, size=4, replace=False).astype(np.uint32)
    centers = cset[rng.integers(0, 4, len(poses))]
    res = small_case["resolution"]
    fallbacks = 0
    for radius, accept in ((om.diameter, 0.0), (om.diameter, 0.02), (om.diameter, 0.5), (2.5 * res, 0.0), (0.0, 0.0)):
        gs.ctx.set_option("debug_ed", 1)
        try:
            got = gs.score_hypotheses_early_drop_ball(gm, poses, centers, radius, DT, accept)
        finally:
            gs.ctx.set_option("debug_ed", 0)
        err = capfd.readouterr().err
        for line in err.splitlines():
            if "rows to the bit-array kernel" in line:
                fallbacks += int(line.split("pending,")[1].split()[0])
        with gs.ctx.option("ed_hist", 0):
            want = gs.score_hypotheses_early_drop_ball(gm, poses, centers, radius, DT, accept)
            with gs.ctx.option("ed_prefix_group", 0):
                want2 = gs.score_hypotheses_early_drop_ball(gm, poses, centers, radius, DT, accept)
        with gs.ctx.option("ed_ball_list", 0):
            want3 = gs.score_hypotheses_early_drop_ball(gm, poses, centers, radius, DT, accept)
        assert all(np.array_equal(x, y) for x, y in zip(got, want3)), (radius, accept)
        assert all(np.array_equal(x, y) for x, y in zip(got, want)), (radius, accept)
        assert all(np.array_equal(x, y) for x, y in zip(got, want2)), (radius, accept)
        for h in list(range(0, len(poses), 7)):
            nn = os_.radius_subset(int(centers[h]), radius)
            sc, mc, score, saved = os_.project(om, poses[h], accept, DT, subset=nn, early_out=True)
            assert got[0][h] == len(sc) and got[3][h] == saved, (radius, accept, h)
            assert abs(got[1][h] - score) <= SCORE_RTOL * abs(score) + SCORE_ATOL_PER_TERM * max(1, len(sc))
    assert fallbacks > 0
    # a centre that is not a finite point has an empty neighbourhood: every hypothesis of it scores nothing
    S2 = small_case["scene"]["surfels"].copy()
    S2[int(cset[0]), 0] = np.nan
    g2 = capi.Scene(gpu_ctx, S2, small_case["scene"]["tangent_mask"])
    got = g2.score_hypotheses_early_drop_ball(gm, poses, centers, om.diameter, DT, 0.0)
    with gpu_ctx.option("ed_hist", 0):
        want = g2.score_hypotheses_early_drop_ball(gm, poses, centers, om.diameter, DT, 0.0)
    assert all(np.array_equal(x, y) for x, y in zip(got, want))
    assert (got[0][centers == cset[0]] == 0).all() and (got[0][centers != cset[0]] > 0).any()
    only = np.full(len(poses), cset[0], np.uint32)  # ... and a batch of nothing but such hypotheses
    got = g2.score_hypotheses_early_drop_ball(gm, poses, only, om.diameter, DT, 0.0)
    assert (got[0] == 0).all() and (got[1] == 0).all()


def test_search_batch_equals_generate_score_scan(gpu_small, oracle_small, small_case):
    """tm_search_batch (one find_in_subset batch on the device) == tm_generate_hypotheses +
    tm_score_hypotheses_[early_drop_]ball + the host scan for the first sufficient hypothesis in visit order
    (reference include/impl/scene.hpp:284-350), for thresholds that put the hit early, late and nowhere."""
    gs, gm = gpu_small["scene"], gpu_small["model"]
    om = oracle_small["model"]
    pairs = sample_pairs(small_case, om, n_first=16, per_first=8, seed=31)
    diam = om.diameter
    mask = np.zeros(gs.n, np.int32)
    mask[np.random.default_rng(8).integers(0, gs.n, 500)] = 1
    hits = 0
    for use_mask in (False, True):
        gs.set_mask(mask if use_mask else None)
        try:
            poses, prov, _ = gs.generate_hypotheses(gm, pairs)  # a masked sample point generates nothing
            centers = pairs[prov[:, 0], 0].astype(np.uint32)
            assert len(poses) > 500
            for early_drop in (True, False):
                if early_drop:
                    c, sc, dr, _ = gs.score_hypotheses_early_drop_ball(gm, poses, centers, diam, DT, 0.5)
                else:
                    c, sc = gs.score_hypotheses_ball(gm, poses, centers, diam, DT)
                    dr = np.full(len(c), -1, np.int32)
                order = np.sort(c[sc > 0])
                for sufficient in (-1.0, float(np.median(order)), float(order[-3]), float(order[-1]) - 0.5,
                                   float(order[-1]), 1e9):
                    ok = np.nonzero((sc > 0) & (c.astype(np.float32) > np.float32(sufficient)))[0]
                    want = int(ok[0]) if len(ok) else -1
                    idx, n_hyp, cnt, dropped, score, pose = gs.search_batch(gm, pairs, diam, DT, 0.5, early_drop,
                                                                            sufficient, min_diameter_factor=0.2)
                    assert n_hyp == len(poses)
                    assert idx == want, (early_drop, sufficient)
                    if want >= 0:
                        hits += 1
                        assert cnt == c[want] and score == sc[want] and dropped == dr[want]
                        assert np.array_equal(pose, poses[want].reshape(-1))
        finally:
            gs.set_mask(None)
    assert hits >= 12
    # the other routes through the early-drop rule give the same record (segment sizes, no segments, the
    # three-kernel rule, one workgroup per pending hypothesis, no prefix kernel)
    poses, prov, _ = gs.generate_hypotheses(gm, pairs)
    centers = pairs[prov[:, 0], 0].astype(np.uint32)
    c, sc, dr, _ = gs.score_hypotheses_early_drop_ball(gm, poses, centers, diam, DT, 0.5)
    order = np.sort(c[sc > 0])
    for env in ({"search_segment": 0}, {"search_segment": 300}, {"ed_unfused": 1}, {"ed_narrow": 1},
                {"ed_one_kernel": 1}, {"ed_wave_rows": 0}, {"ed_wave_rows": 1}, {"ed_prefix_list": 0},
                {"ed_wave_rows": 3, "search_segment": 300}, {"ed_hist": 0}, {"ed_hist": 0, "ed_wave_rows": 0}, {"ed_prefix_group": 0}, {"ed_prefix_group": 0, "ed_hist": 0}, {"ed_ball_list": 0}, {"ed_prefix_group": 5},
                {"ed_hist": 0, "search_segment": 300}, {"search_step_check": 0}, {"search_step_check": 0, "search_segment": 300}):
        old_env = {k: gs.ctx.get_option(k) for k in env}
        for k, v in env.items():
            gs.ctx.set_option(k, v)
        try:
            for sufficient in (float(np.median(order)), float(order[-2]), float(order[-1])):
                ok = np.nonzero((sc > 0) & (c.astype(np.float32) > np.float32(sufficient)))[0]
                want = int(ok[0]) if len(ok) else -1
                idx, n_hyp, cnt, dropped, score, pose = gs.search_batch(gm, pairs, diam, DT, 0.5, True, sufficient,
                                                                        min_diameter_factor=0.2)
                assert idx == want and n_hyp == len(poses), (env, sufficient)
                if want >= 0:
                    assert cnt == c[want] and score == sc[want] and dropped == dr[want], (env, sufficient)
        finally:
            for k, v in old_env.items():
                gs.ctx.set_option(k, v)
    # several batches in one call (the chunks of a round) == one call per batch, also with an empty batch
    offs = np.array([0, 7, 7, 30, 64, len(pairs)], np.uint32)
    for early_drop in (True, False):
        for sufficient in (float(np.median(order)), float(order[-2]), 1e9):
            many = gs.search_batches(gm, pairs, offs, diam, DT, 0.5, early_drop, sufficient, min_diameter_factor=0.2)
            assert len(many) == len(offs) - 1
            for b, got in enumerate(many):
                one = gs.search_batch(gm, pairs[offs[b]:offs[b + 1]], diam, DT, 0.5, early_drop, sufficient,
                                      min_diameter_factor=0.2)
                assert got[:5] == one[:5], (early_drop, sufficient, b)
                if one[0] >= 0:
                    assert np.array_equal(got[5], one[5])
            assert many[1][0] == -1 and many[1][1] == 0
            # the batches' segment steps merged into shared launches (default) == one batch after the other
            with gs.ctx.option("search_per_range", 1):
                seq = gs.search_batches(gm, pairs, offs, diam, DT, 0.5, early_drop, sufficient, min_diameter_factor=0.2)
            for got, want_ in zip(many, seq):
                assert got[:5] == want_[:5] and (got[0] < 0 or np.array_equal(got[5], want_[5]))
            with gs.ctx.option("search_step_check", 0):  # every step queued, no look at the stop words in between
                for got, want_ in zip(many, gs.search_batches(gm, pairs, offs, diam, DT, 0.5, early_drop, sufficient,
                                                              min_diameter_factor=0.2)):
                    assert got[:5] == want_[:5] and (got[0] < 0 or np.array_equal(got[5], want_[5]))
            with gs.ctx.option("search_segment", 300):  # several steps per batch, batches of unequal length
                short = gs.search_batches(gm, pairs, offs, diam, DT, 0.5, early_drop, sufficient, min_diameter_factor=0.2)
                with gs.ctx.option("ed_wave_rows", 1):  # the continuation one pending hypothesis per batch first
                    short1 = gs.search_batches(gm, pairs, offs, diam, DT, 0.5, early_drop, sufficient,
                                               min_diameter_factor=0.2)
            with gs.ctx.option("ed_wave_rows", 0):  # ... or all pending hypotheses at once
                nowave = gs.search_batches(gm, pairs, offs, diam, DT, 0.5, early_drop, sufficient, min_diameter_factor=0.2)
            for other in (short, short1, nowave):
                for got, want_ in zip(many, other):
                    assert got[:5] == want_[:5] and (got[0] < 0 or np.array_equal(got[5], want_[5]))
    # degenerate batches
    assert gs.search_batch(gm, np.zeros((0, 2), np.uint32), diam, DT, 0.5, True, 0.0)[0] == -1
    with pytest.raises(capi.TmError):
        gs.search_batch(gm, np.array([[0, gs.n]], np.uint32), diam, DT, 0.5, True, 0.0)


def test_search_batches_with_a_long_pending_list(gpu_small, oracle_small, small_case):
    """A merged segment step with more pending hypotheses than k_pending_sort's LDS rank sort holds (4096): with
    accept_prob 0 nothing drops at the first checkpoint, so every hypothesis of the step is pending. The rows must keep
    their batch (each batch follows ITS stop word, reference include/impl/scene.hpp:330-338 per chunk) and their visit
    order: the merged call == one batch after the other, for thresholds that stop batches at different places."""
    gs, gm = gpu_small["scene"], gpu_small["model"]
    om = oracle_small["model"]
    pairs = sample_pairs(small_case, om, n_first=96, per_first=24, seed=77)
    diam = om.diameter
    poses, prov, _ = gs.generate_hypotheses(gm, pairs)
    # batches of unequal length over the sample pairs; hypothesis offsets from the provenance rows
    cuts = np.array([0, 1, 2, 4, 7, 11, 16, 24], np.float64) / 24.0
    offs = np.unique((cuts * len(pairs)).astype(np.uint32))
    offs[-1] = len(pairs)
    per_pair = np.bincount(prov[:, 0], minlength=len(pairs))
    hyp_starts = np.concatenate([[0], np.cumsum(per_pair)])[offs]
    in_first_step = np.minimum(np.diff(hyp_starts), 2048).sum()
    assert in_first_step > 4096 + 512, in_first_step  # the long-list route of k_pending_sort runs
    centers = pairs[prov[:, 0], 0].astype(np.uint32)
    c, sc, dr, _ = gs.score_hypotheses_early_drop_ball(gm, poses, centers, diam, DT, 0.0)
    assert (dr < 0).all()  # nothing dropped: every hypothesis went through the continuation
    order = np.sort(c[sc > 0])
    differing_stops = 0
    for sufficient in (float(order[len(order) // 2]), float(order[-40]), float(order[-3]), 1e9):
        many = gs.search_batches(gm, pairs, offs, diam, DT, 0.0, True, sufficient, min_diameter_factor=0.2)
        with gs.ctx.option("search_per_range", 1):
            seq = gs.search_batches(gm, pairs, offs, diam, DT, 0.0, True, sufficient, min_diameter_factor=0.2)
        with gs.ctx.option("ed_wave_rows", 0):
            nowave = gs.search_batches(gm, pairs, offs, diam, DT, 0.0, True, sufficient, min_diameter_factor=0.2)
        for b, (got, want_, got2) in enumerate(zip(many, seq, nowave)):
            # the host scan over the whole-batch scores says the same
            h0, h1 = int(hyp_starts[b]), int(hyp_starts[b + 1])
            ok = np.nonzero((sc[h0:h1] > 0) & (c[h0:h1].astype(np.float32) > np.float32(sufficient)))[0]
            assert want_[0] == (int(ok[0]) if len(ok) else -1), (sufficient, b)
            assert got[:5] == want_[:5] and got2[:5] == want_[:5], (sufficient, b)
            if want_[0] >= 0:
                assert np.array_equal(got[5], want_[5])
        found = [r[0] >= 0 for r in many]
        differing_stops += int(any(found) and not all(found))
    assert differing_stops >= 1  # some batch had its hypothesis while another was still looking


def test_extract_batch_equals_single_calls(gpu_small, small_case):
    """tm_extract_correspondences_batch (finish_find for several poses, one read-back) == one call per pose"""
    gm, gs = gpu_small["model"], gpu_small["scene"]
    poses = random_poses_near_truth(small_case, 7, seed=3)
    got = gs.extract_correspondences_batch(gm, poses, DT)
    hits = 0
    for k in range(len(poses)):
        sc, mc, score = gs.extract_correspondences(gm, poses[k], DT)
        assert np.array_equal(got[k][0], sc) and np.array_equal(got[k][1], mc) and got[k][2] == score
        hits += int(len(sc) > 100)
    assert hits >= 2


def _arbitrate_numpy(lists, n_scene, tmask, min_points, min_points_all):
    """the greedy accept of find_parallel (reference include/impl/scene.hpp:193-237) in plain numpy: order by list size
    (descending, stable); a candidate keeps the points no ACCEPTED earlier candidate holds; accepted when enough of them
    are tangent points and enough in all. Returns (verdict, kept, tangent_kept, needs_trimming)."""
    order = sorted(range(len(lists)), key=lambda k: -len(lists[k]))
    used = np.zeros(n_scene, bool)
    verdict = np.zeros(len(lists), np.int32)
    kept = np.zeros(len(lists), np.uint32)
    tan = np.zeros(len(lists), np.uint32)
    trimmed = False
    for k in order:
        idx = lists[k].astype(np.int64)
        free = idx[~used[idx]]
        kept[k], tan[k] = len(free), int(np.count_nonzero(tmask[free]))
        if tan[k] >= min_points and kept[k] >= min_points_all:
            verdict[k] = 1
            used[free] = True
            trimmed = trimmed or len(free) != len(idx)
    return verdict, kept, tan, trimmed


def test_extract_arbitrate_batch_equals_host_arbitration(gpu_small, small_case):
    """tm_extract_arbitrate_batch: finish_find of several poses + the arbitration of find_parallel where the lists are.
    Verdicts and both counts per candidate equal the greedy pass in numpy over the lists of the plain batch call — for
    disjoint instances, for near-duplicates of an instance (rejected: an accepted one holds their points), for thresholds
    that reject an early candidate whose points a later one then takes — and -1 for all when an accepted candidate would
    have to be trimmed. tm_scene_mask_add_extracted then masks exactly the accepted rows' points."""
    gm, gs = gpu_small["model"], gpu_small["scene"]
    tmask = small_case["scene"]["tangent_mask"].astype(bool)
    near = random_poses_near_truth(small_case, 6, seed=11)          # several poses per planted instance: overlaps
    far = random_poses_near_truth(small_case, 3, seed=12, jitter=0.5)  # poor poses: short lists
    poses = np.concatenate([np.asarray(near).reshape(-1, 16), np.asarray(far).reshape(-1, 16)])
    plain = gs.extract_correspondences_batch(gm, poses, DT)
    lists = [p[0] for p in plain]
    sizes = sorted(len(l) for l in lists)
    decided_cases = 0
    for min_points, min_all in ((1, 1), (1, sizes[len(sizes) // 2]), (1, sizes[-1]), (10**9, 1), (1, sizes[-1] + 1),
                                (max(1, int(np.count_nonzero(tmask)) // 50), sizes[len(sizes) // 3])):
        want_v, want_k, want_t, trimmed = _arbitrate_numpy(lists, gs.n, tmask, min_points, min_all)
        v, k, t, counts, got = gs.extract_arbitrate_batch(gm, poses, DT, min_points, min_all)
        assert np.array_equal(counts, [len(l) for l in lists])
        if trimmed:
            assert (v == -1).all(), (min_points, min_all)  # the host has to trim an accepted candidate's lists
            for g, p in zip(got, plain):
                assert np.array_equal(g[0], p[0]) and np.array_equal(g[1], p[1])  # every row written, as by the plain call
            continue
        decided_cases += 1
        assert np.array_equal(v, want_v), (min_points, min_all, v, want_v)
        assert np.array_equal(k, want_k) and np.array_equal(t, want_t)
        for j in range(len(poses)):
            if v[j] == 1:
                assert np.array_equal(got[j][0], plain[j][0]) and np.array_equal(got[j][1], plain[j][1]) and got[j][2] == plain[j][2]
        # the accepted rows' points, and nothing else, reach mask_
        acc = np.flatnonzero(v == 1).astype(np.uint32)
        finite = np.isfinite(small_case["scene"]["surfels"][:, 0:3]).all(axis=1)
        try:
            gs.set_mask(None)
            gs.mask_add_extracted(acc)
            masked = np.zeros(gs.n, bool)
            for j in acc:
                masked[plain[j][0]] = True
            assert gs.live_count() == int(np.count_nonzero(finite & ~masked))
        finally:
            gs.set_mask(None)
    assert decided_cases >= 3


def test_radius_lists_equal_single_queries(gpu_small, oracle_small, small_case):
    gs, os_ = gpu_small["scene"], oracle_small["scene"]
    diam = oracle_small["model"].diameter
    rng = np.random.default_rng(23)
    centers = rng.integers(0, gs.n, 11).astype(np.uint32)
    for r in (diam, 0.0):
        lists = gs.radius_lists(centers, r)
        for k, c in enumerate(centers):
            assert np.array_equal(lists[k], os_.radius_subset(int(c), r))
        # the sampling loop's view of the same neighbourhoods: size + (rank, index) of the tangent members
        counts, tangent = gs.radius_tangent_lists(centers, r)
        tm = small_case["scene"]["tangent_mask"]
        for k in range(len(centers)):
            assert counts[k] == len(lists[k])
            want = np.array([(rk, j) for rk, j in enumerate(lists[k]) if tm[j]], np.uint32).reshape(-1, 2)
            assert np.array_equal(tangent[k], want), (r, k)  # ascending index order, ranks in the full list
    assert sum(len(t) for t in tangent) >= 0
    counts, tangent = gs.radius_tangent_lists(centers, diam)
    assert sum(len(t) for t in tangent) > 100
    # a few entries of a neighbourhood without its list (the traits search's draws)
    full = os_.radius_subset(int(centers[0]), diam)
    ranks = np.array([0, 1, len(full) // 2, len(full) - 1, 7, 7], np.uint32)
    n, picked = gs.radius_count_and_select(int(centers[0]), diam, ranks)
    assert n == len(full) and np.array_equal(picked, full[ranks])
    with pytest.raises(capi.TmError):
        gs.radius_count_and_select(int(centers[0]), diam, np.array([len(full)], np.uint32))


def test_grouped_scoring_edge_sizes(gpu_ctx, gpu_small, oracle_small, small_case):
    """The grouped kernel (16 hypotheses share the scene chunks they reach) against the exhaustive kernel
    and the oracle at the sizes where its bookkeeping has edges: fewer hypotheses than a group, one more than a
    group, a last partial group; a scene smaller than one 512-point chunk; a masked scene."""
    g, s = gpu_small["model"], gpu_small["scene"]
    poses = random_poses_near_truth(small_case, 100, seed=23)
    cb, sb = s.score_hypotheses(g, poses, DT, mode=capi.SCORE_BRUTEFORCE)
    co, so, _ = oracle_small["scene"].score_hypotheses(oracle_small["model"], poses, 0.5, DT, threads=8)
    assert np.array_equal(cb, co)
    for n in (1, 15, 16, 17, 31, 32, 33, 100):
        cg, sg = s.score_hypotheses(g, poses[:n], DT, mode=capi.SCORE_CULLED)
        assert np.array_equal(cg, cb[:n]), n
        assert np.array_equal(sg, sb[:n]), n  # fixed-point sums: identical, not merely close
    # the items binned by the model region they land in (default for large batches only): same sums in any order
    for shift in (0, 3, 4, 6):
        with gpu_ctx.option("item_sort", shift):
            cg, sg = s.score_hypotheses(g, poses, DT, mode=capi.SCORE_CULLED)
        assert np.array_equal(cg, cb) and np.array_equal(sg, sb), shift
    # the other instantiations and hand-out forms of the item kernel: 64-bit addresses for its gathers (what a voxel table
    # beyond 4 GiB takes), a fixed item share per wave, no continuation in other XCDs' items, short runs per XCD, a small
    # persistent grid — same integers every way
    for opts in ({"score_off32": 0}, {"item_sort": 4},
                 {"score_item_wgs": 8}, {"score_off32": 0, "item_sort": 3}):
        old = {k: gpu_ctx.get_option(k) for k in opts}
        for k, v in opts.items():
            gpu_ctx.set_option(k, v)
        try:
            cg, sg = s.score_hypotheses(g, poses, DT, mode=capi.SCORE_CULLED)
        finally:
            for k, v in old.items():
                gpu_ctx.set_option(k, v)
        assert np.array_equal(cg, cb) and np.array_equal(sg, sb), opts
    # the per-hypothesis kernel of round 1 (still used for subset scoring) agrees as well
    with gpu_ctx.option("score_grouped", 0):
        cg, sg = s.score_hypotheses(g, poses, DT, mode=capi.SCORE_CULLED)
    assert np.array_equal(cg, cb) and np.array_equal(sg, sb)
    # a slot budget so small that the batch goes through in several launches of whole groups (what a 1M-hypothesis
    # batch on a 10M-point scene does with the default 256 MiB): 1 MiB = 131,072 slots; the 20k-point scene has 40
    # chunks, 3000 hypotheses = 188 groups x 40 = 7,520 slots, so shrink the scene side instead: many hypotheses
    many = np.tile(poses, (40, 1))[:3990]
    cm1, sm1 = s.score_hypotheses(g, many, DT, mode=capi.SCORE_CULLED)
    gpu_ctx.set_option("score_slot_mib", 1)
    try:
        import triplet_match_amd.capi as _c
        assert gpu_ctx.get_option("score_slot_mib") == 1
        big = np.tile(poses, (800, 1))[:70000]  # 4,375 groups x 40 chunks = 175,000 slots > 131,072: two launches
        cbig, sbig = s.score_hypotheses(g, big, DT, mode=capi.SCORE_CULLED)
    finally:
        gpu_ctx.set_option("score_slot_mib", 256)
    assert np.array_equal(cbig[:3990], cm1) and np.array_equal(sbig[:3990], sm1)
    assert np.array_equal(cbig[:100], cb) and np.array_equal(cbig[69900:70000], cb[:100])
    # unknown switches and the timing experiments are refused by the product library
    with pytest.raises(capi.TmError):
        gpu_ctx.set_option("no_such_switch", 1)
    with pytest.raises(capi.TmError) as e:
        gpu_ctx.set_option("ablate", 1)
    assert e.value.code == capi.ERR_UNSUPPORTED and b"timing" not in capi.lib().tm_version()
    # the environment is read once, in tm_ctx_create: a leftover TM_ABLATE is ignored with a warning by the product
    # library (it used to fail every context creation), a value out of range is an error and leaves nothing behind
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from triplet_match_amd import capi\n"
            "try:\n"
            "    c = capi.Context(0)\n"
            "    print('created', c.get_option('ablate'), c.get_option('item_sort'))\n"
            "except capi.TmError as e:\n"
            "    print('refused', e.code)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TM_ABLATE="1", TM_ITEM_SORT="3"))
    assert r.returncode == 0 and r.stdout.split() == ["created", "0", "3"] and "TM_ABLATE=1 ignored" in r.stderr, r
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, TM_ITEM_SORT="17"))
    assert r.returncode == 0 and r.stdout.split() == ["refused", str(capi.ERR_INVALID)], r
    # a scene of 300 points: less than one chunk
    sc = small_case["scene"]
    inst = np.nonzero(sc["instance_id"] == 0)[0][:300]
    tiny = capi.Scene(gpu_ctx, sc["surfels"][inst], sc["tangent_mask"][inst])
    ct, st = tiny.score_hypotheses(g, poses, DT, mode=capi.SCORE_CULLED)
    ctb, stb = tiny.score_hypotheses(g, poses, DT, mode=capi.SCORE_BRUTEFORCE)
    assert np.array_equal(ct, ctb) and np.array_equal(st, stb) and ct.max() > 100
    # a mask over half of the scene
    mask = (np.arange(len(sc["surfels"])) % 2).astype(np.int32)
    s.set_mask(mask)
    try:
        cm, sm = s.score_hypotheses(g, poses, DT, mode=capi.SCORE_CULLED)
        cmb, smb = s.score_hypotheses(g, poses, DT, mode=capi.SCORE_BRUTEFORCE)
        assert np.array_equal(cm, cmb) and np.array_equal(sm, smb) and cm.max() < cb.max()
    finally:
        s.set_mask(None)


@pytest.fixture(params=[0, 1], ids=["tree_on_device", "tree_on_host"])
def cloud_tree_route(request, gpu_ctx):
    """tm_cloud's tree built on the device (curve order + linear BVH, the default) or on the host (median kd-tree)"""
    old = gpu_ctx.get_option("model_host_build")
    gpu_ctx.set_option("model_host_build", request.param)
    yield request.param
    gpu_ctx.set_option("model_host_build", old)


def test_cloud_neighbour_queries_equal_oracle_and_scipy(gpu_ctx, small_case, cloud_tree_route, curv_walk):
    """tm_cloud (the neighbour accessors of pointcloud<Point>, include/pointcloud:38-63): kNN (kd-tree route k <= 32,
    one-pass route k > 32) and radius search against the oracle's exhaustive scan — indices and squared distances
    bit for bit, order (d2, index) — and, independently of both, against scipy's cKDTree (same neighbour SETS up to
    ties at the last distance). Includes duplicated points (ties), a non-finite point, a subset tree, queries that
    are not cloud points, and curvature over an explicit neighbour list."""
    from oracle import tmo
    from scipy.spatial import cKDTree
    cloud = small_case["scene"]["surfels"][:6000].copy()
    cloud[700:720, 0:3] = cloud[100:120, 0:3]  # exact duplicates: ties broken by index
    cloud[33, 0] = np.nan
    gc = capi.Cloud(gpu_ctx, cloud)
    assert gc.tree_size() == 5999
    rng = np.random.default_rng(5)
    qidx = np.concatenate([[0, 100, 700, 5999], rng.integers(0, 6000, 60)])
    queries = cloud[qidx, 0:3].copy()
    queries[-10:] += rng.normal(size=(10, 3)).astype(np.float32) * 0.01  # not cloud points
    queries = queries[np.isfinite(queries).all(axis=1)]
    finite = np.isfinite(cloud[:, 0:3]).all(axis=1)
    tree = cKDTree(cloud[finite, 0:3].astype(np.float64))
    orig = np.nonzero(finite)[0]
    for k in (1, 2, 30, 32, 33, 100):
        idx, d2, cnt = gc.knn(queries, k)
        assert (cnt == k).all()
        for t, q in enumerate(queries):
            oi, od = tmo.cloud_knn(cloud, q, k)
            assert np.array_equal(idx[t], oi) and np.array_equal(d2[t].view(np.uint32), od.view(np.uint32)), (k, t)
        # scipy: same distances (f64 vs f32: compare sorted distances loosely, sets exactly where no tie at the edge)
        sd, si = tree.query(queries.astype(np.float64), k=k)
        sd = sd.reshape(len(queries), k)
        assert np.allclose(np.sqrt(d2.astype(np.float64)), sd, rtol=1e-5, atol=1e-6)
    r = 6.0 * small_case["resolution"]
    for t, q in enumerate(queries[:20]):
        gi, gd = gc.radius_search(q, r)
        oi, od = tmo.cloud_radius_search(cloud, q, r)
        assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
        assert (np.diff(gd) >= 0).all() and (gd <= np.float32(r) * np.float32(r)).all()
        si = orig[np.array(tree.query_ball_point(q.astype(np.float64), r * (1 - 1e-6)), dtype=np.int64)]
        assert set(si.tolist()) <= set(gi.tolist())  # everything clearly inside is there
        si_out = orig[np.array(tree.query_ball_point(q.astype(np.float64), r * (1 + 1e-6)), dtype=np.int64)]
        assert set(gi.tolist()) <= set(si_out.tolist())  # nothing clearly outside is
        # curvature over that neighbour list == the oracle's principal_curvatures on the same list
        if len(gi) >= 3:
            p_idx = int(gi[0])
            assert np.array_equal(gc.curvature_of(p_idx, gi).view(np.uint32),
                                  tmo.cloud_curvature_of(cloud, p_idx, gi).view(np.uint32))
    # a non-finite query has no neighbours; radius 0 around a cloud point returns its duplicates
    idx, d2, cnt = gc.knn(np.array([[np.inf, 0, 0]], np.float32), 4)
    assert cnt[0] == 0 and (idx == -1).all()
    gi, gd = gc.radius_search(cloud[100, 0:3], 0.0)
    assert gi.tolist() == [100, 700] and (gd == 0).all()
    # set_indices: a tree over a subset
    sub = np.arange(0, 6000, 3, dtype=np.uint32)
    gs = capi.Cloud(gpu_ctx, cloud, subset=sub)
    assert gs.tree_size() == len(sub) - 1  # point 33 is in the subset and not finite
    idx, d2, cnt = gs.knn(queries[:16], 8)
    for t, q in enumerate(queries[:16]):
        oi, od = tmo.cloud_knn(cloud, q, 8, subset=sub)
        assert np.array_equal(idx[t], oi) and np.array_equal(d2[t].view(np.uint32), od.view(np.uint32))
    # fewer points than k: padded rows
    tiny = capi.Cloud(gpu_ctx, cloud[:5])
    idx, d2, cnt = tiny.knn(cloud[:2, 0:3], 9)
    assert (cnt == 5).all() and (idx[:, 5:] == -1).all() and np.isinf(d2[:, 5:]).all()


def _brute_knn(points, queries, k):
    """k smallest (d2, index) per query with d2 = (dx*dx + dy*dy) + dz*dz in float32, as the kernels form it"""
    P = points.astype(np.float32)
    idx = np.full((len(queries), k), -1, np.int32)
    d2 = np.full((len(queries), k), np.inf, np.float32)
    for t, q in enumerate(queries.astype(np.float32)):
        d = P - q
        v = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        order = np.lexsort((np.arange(len(P)), v))[:k]
        idx[t, :len(order)] = order
        d2[t, :len(order)] = v[order]
    return idx, d2


def test_neighbour_lists_with_many_ties(gpu_ctx, cloud_tree_route, curv_walk):
    """clouds made of ties: an integer lattice (dozens of points at every distance), the same point a hundred times,
    collinear points, a lattice stored twice — the k nearest come out in (d2, index) order on every route (tree on the
    device / host, eight lanes per query / one thread per query), equal to a numpy scan of all points; k from 1 to the
    list's capacity, more neighbours asked than there are points, queries on and off the lattice"""
    rng = np.random.default_rng(21)
    g = np.arange(9, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    lattice = lattice[rng.permutation(len(lattice))]
    clouds = {
        "lattice": lattice,
        "lattice twice": np.concatenate([lattice, lattice[::-1]]),
        "one point": np.tile(np.array([[0.5, -1.0, 2.0]], np.float32), (100, 1)),
        "collinear": np.stack([np.arange(300, dtype=np.float32) % 50, np.zeros(300, np.float32), np.zeros(300, np.float32)], 1),
        "few": lattice[:5],
    }
    for name, pts in clouds.items():
        surfels = np.zeros((len(pts), 12), np.float32)
        surfels[:, 0:3] = pts
        surfels[:, 3] = 1.0
        surfels[:, 6] = 1.0
        gc = capi.Cloud(gpu_ctx, surfels)
        queries = np.concatenate([pts[rng.integers(0, len(pts), 24)], pts[:8] + np.float32(0.5),
                                  (rng.random((8, 3)) * 8).astype(np.float32)])
        for k in (1, 2, 7, 27, 30, 32):
            idx, d2, cnt = gc.knn(queries, k)
            bi, bd = _brute_knn(pts, queries, k)
            assert np.array_equal(idx, bi), (name, k)
            assert np.array_equal(d2.view(np.uint32), bd.view(np.uint32)), (name, k)
            assert (cnt == min(k, len(pts))).all()
        # the curvature pass walks the same lists (self included): its rows agree between the routes by the other tests;
        # here the lists' k-th distance is what cloud->resolution() reads with k = 2 — zero wherever a point has a twin
        if len(pts) >= 2:
            res, nn = gpu_ctx.cloud_resolution(surfels)
            bi, bd = _brute_knn(pts, pts, 2)
            assert np.array_equal(nn.view(np.uint32), np.sqrt(bd[:, 1]).astype(np.float32).view(np.uint32)), name


def test_gpu_curvature_and_resolution_against_scipy_and_numpy(gpu_ctx, curv_walk):
    """tm_cloud_curvature / tm_cloud_resolution (kd-tree kNN kernel, own Jacobi) against the float64 scipy cKDTree +
    numpy.linalg.eigh vectors of tests/golden/curvature_20k.npz — an answer that passes through none of our code"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "curvature_20k.npz"))
    cloud, _ = synth.synth_model(20000, seed=4)
    c = gpu_ctx.cloud_curvature(cloud, int(g["k"]))
    assert np.abs(c[:, 3] - g["pc_min"]).max() < 1e-6 and np.abs(c[:, 4] - g["pc_max"]).max() < 1e-6
    res, nn1 = gpu_ctx.cloud_resolution(cloud)
    assert (np.abs(nn1 - g["nn1"]) <= 1e-6 * g["nn1"]).all()
    assert abs(res - float(np.mean(g["nn1"].astype(np.float64)))) < 1e-5 * res
    # the neighbour accessor ends at the same 30th distance as scipy's tree
    gc = capi.Cloud(gpu_ctx, cloud)
    idx, d2, cnt = gc.knn(cloud[:2000, 0:3], int(g["k"]))
    assert (np.abs(d2[:, -1] - g["kth_d2"][:2000]) <= 2e-6 * g["kth_d2"][:2000] + 1e-12).all()


def test_curvature_subset_equals_rows_of_the_full_call(gpu_ctx, small_case, curv_walk):
    """tm_cloud_curvature_subset (the points that pass the tangent test only) == those rows of tm_cloud_curvature,
    bit for bit; non-finite listed points keep zero rows"""
    cloud = small_case["scene"]["surfels"][:8000].copy()
    cloud[77, 1] = np.inf
    full = gpu_ctx.cloud_curvature(cloud, 30)
    idx = np.concatenate([[77, 0, 7999], np.nonzero(np.linalg.norm(cloud[:, 9:12], axis=1) > 0.7)[0]]).astype(np.uint32)
    sub = gpu_ctx.cloud_curvature_subset(cloud, idx, 30)
    assert np.array_equal(sub.view(np.uint32), full[idx].view(np.uint32)) and (sub[0] == 0).all()
    assert len(idx) > 100


def test_scene_curvature_and_late_tangent_mask(gpu_ctx, oracle_small, small_case, curv_walk):
    """tm_scene_curvature_subset (tree built on the device over the scene's curve order) == tm_cloud_curvature_subset
    (host-built kd-tree) == the oracle, bit for bit, with a mask set and non-finite points in the cloud; a scene
    created with an empty tangent mask and given tangent_mask_ afterwards (tm_scene_set_tangent_mask) == a scene
    created with it."""
    S = small_case["scene"]["surfels"].copy()
    tmk = small_case["scene"]["tangent_mask"]
    S[123, 0] = np.nan
    S[4567, 2] = np.inf
    idx = np.concatenate([[123, 4567, 0, len(S) - 1], np.nonzero(tmk)[0]]).astype(np.uint32)
    late = capi.Scene(gpu_ctx, S, np.zeros(len(S), np.int32))
    mask = np.zeros(len(S), np.int32)
    mask[::3] = 1
    late.set_mask(mask)  # mask_ does not take points out of the neighbourhoods
    got = late.curvature_subset(idx, 30)
    want = gpu_ctx.cloud_curvature_subset(S, idx, 30)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (got[0] == 0).all() and (got[1] == 0).all() and (got[2:] != 0).any()
    ref = tmo.cloud_curvature(S, 30)[idx]
    assert np.array_equal(got[2:].view(np.uint32), ref[2:].view(np.uint32))
    for k in (1, 7, 32):
        assert np.array_equal(late.curvature_subset(idx[:200], k).view(np.uint32),
                              gpu_ctx.cloud_curvature_subset(S, idx[:200], k).view(np.uint32))
    # tiny clouds: fewer points than k, a single leaf
    for n in (1, 5, 9, 40):
        tiny = capi.Scene(gpu_ctx, S[1000:1000 + n], np.zeros(n, np.int32))
        ids = np.arange(n, dtype=np.uint32)
        assert np.array_equal(tiny.curvature_subset(ids, 30).view(np.uint32),
                              gpu_ctx.cloud_curvature_subset(S[1000:1000 + n], ids, 30).view(np.uint32))
    late.set_mask(None)
    late.set_tangent_mask(tmk)
    early = capi.Scene(gpu_ctx, S, tmk)
    gm = capi.Model(gpu_ctx, small_case["model"], small_case["resolution"])
    gm.build_voxels()
    gm.build_pairs(small_case["subset"])
    poses = random_poses_near_truth(small_case, 24)
    a, b = late.score_hypotheses(gm, poses, DT), early.score_hypotheses(gm, poses, DT)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[0].max() > 100
    pairs = sample_pairs(small_case, oracle_small["model"], n_first=6, per_first=6, seed=3)
    pa, pb = late.generate_hypotheses(gm, pairs), early.generate_hypotheses(gm, pairs)
    assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1]) and len(pa[0]) > 10
    ta, tb = late.tangent_points(), early.tangent_points()
    assert np.array_equal(ta[0], tb[0]) and np.array_equal(ta[1].view(np.uint32), tb[1].view(np.uint32))
    # tangent_mask_ built on the device in one call (tm_scene_auto_tangent_mask) == the reference's loop
    # (include/impl/scene.hpp:46-54) over the oracle's curvature, with and without the `ok` flags and the curvature test
    T = S[:, 9:12]
    nrm = np.sqrt(((T[:, 0] * T[:, 0] + T[:, 1] * T[:, 1]) + T[:, 2] * T[:, 2]).astype(np.float32))
    curv = tmo.cloud_curvature(S, 30)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio_ok = (curv[:, 3] / curv[:, 4]) < np.float32(0.2)
    ok = (np.arange(len(S)) % 5 != 0).astype(np.uint8)
    for use_ok, k in ((False, 30), (True, 30), (False, 0), (True, 7)):
        cand_want = (nrm > np.float32(0.7)) & ((ok != 0) if use_ok else True)
        if k == 30:
            want_mask = cand_want & ratio_ok
        elif k == 0:
            want_mask = cand_want
        else:
            rows_k = gpu_ctx.cloud_curvature_subset(S, np.nonzero(cand_want)[0].astype(np.uint32), k)
            with np.errstate(divide="ignore", invalid="ignore"):
                rk = (rows_k[:, 3] / rows_k[:, 4]) < np.float32(0.2)
            want_mask = cand_want.copy()
            want_mask[np.nonzero(cand_want)[0]] = rk
        auto = capi.Scene(gpu_ctx, S, np.zeros(len(S), np.int32))
        got_mask, cand, rows = auto.auto_tangent_mask(k, 0.7, 0.2, ok if use_ok else None)
        assert np.array_equal(cand, np.nonzero(cand_want)[0]), (use_ok, k)
        assert np.array_equal(got_mask != 0, want_mask), (use_ok, k)
        if k == 30:
            assert np.array_equal(rows.view(np.uint32), curv[cand].view(np.uint32))
        assert want_mask.sum() > 50
        # ... and the scene behaves as one created with that mask
        twin = capi.Scene(gpu_ctx, S, want_mask.astype(np.int32))
        a, b = auto.score_hypotheses(gm, poses, DT), twin.score_hypotheses(gm, poses, DT)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        ta, tb = auto.tangent_points(), twin.tangent_points()
        assert np.array_equal(ta[0], tb[0]) and np.array_equal(ta[1].view(np.uint32), tb[1].view(np.uint32))
        assert np.array_equal(auto.download()[1], want_mask.astype(np.int32))
    # room for fewer candidates than there are: the count comes back with TM_ERR_CAPACITY
    nc = C.c_uint32(0)
    small = np.zeros(4, np.uint32)
    rc = capi.lib().tm_scene_auto_tangent_mask(auto.h, 30, 0.7, 0.2, None, None, small.ctypes.data_as(capi.u32p), None, 4,
                                               C.byref(nc), None)
    assert rc == capi.ERR_CAPACITY and nc.value == int((nrm > np.float32(0.7)).sum())


def test_concurrent_calls_on_one_context(gpu_small, small_case):
    """thread-safe per handle. The reference's workers call model::query / voxel_query and score from
    several threads at once (include/impl/scene.hpp:146-166,304); calls on one context serialise on its mutex, so
    four threads that score different batches, query the table and look up voxels concurrently (ctypes releases the GIL
    for the duration of a call) get what they get one after the other."""
    import threading
    gs, gm = gpu_small["scene"], gpu_small["model"]
    batches = [random_poses_near_truth(small_case, 96, seed=40 + t) for t in range(4)]
    want = [gs.score_hypotheses(gm, b, DT) for b in batches]
    pts = small_case["model"][::97, :4].astype(np.float32).copy()
    want_vox = [gm.voxel_query(p) for p in pts]
    got = [None] * 4
    vox = [None] * 4
    errors = []

    def work(t):
        try:
            for _ in range(6):
                got[t] = gs.score_hypotheses(gm, batches[t], DT)
                vox[t] = [gm.voxel_query(p) for p in pts]
                gs.radius_subset(int(t), float(gm.diameter))
        except Exception as e:  # noqa: BLE001 - reported below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(4):
        assert np.array_equal(got[t][0], want[t][0]) and np.array_equal(got[t][1], want[t][1])
        assert vox[t] == want_vox


def test_block_cache_of_a_context(small_case):
    """The context's cache of released device blocks (tm_ctx_pool_stats, option pool_mib): scenes created and destroyed in
    a loop are served from the cache after the first round, the cache never exceeds its budget, a smaller budget trims
    at once, 0 switches it off — and the results do not depend on any of it."""
    S, tmk = small_case["scene"]["surfels"], small_case["scene"]["tangent_mask"]
    ctx = capi.Context(0)
    gm = capi.Model(ctx, small_case["model"], small_case["resolution"])
    gm.build_voxels()
    poses = random_poses_near_truth(small_case, 40, seed=77)

    def one_round():
        s = capi.Scene(ctx, S, tmk)
        out = s.score_hypotheses(gm, poses, DT)
        s.close()
        return out
    want = one_round()
    cached0, reused0, alloc0 = ctx.pool_stats()
    assert cached0 > 0 and ctx.get_option("pool_mib") == 4096
    for _ in range(3):
        got = one_round()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    cached1, reused1, alloc1 = ctx.pool_stats()
    assert reused1 - reused0 >= 3 * 10       # a scene is some forty blocks; they come out of the cache ...
    assert alloc1 - alloc0 <= 2              # ... not from hipMalloc
    assert cached1 <= 1024 << 20
    ctx.set_option("pool_mib", 1)            # a smaller budget takes effect at once
    assert ctx.pool_stats()[0] <= 1 << 20
    ctx.set_option("pool_mib", 0)            # off: nothing is kept, every block goes back to the device
    got = one_round()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert ctx.pool_stats()[0] == 0
    gm.close()


def test_scene_from_device_buffers(gpu_ctx, gpu_small, small_case):
    """tm_scene_create_dev: a cloud (and tangent_mask_) already on the device make the same scene as the host call —
    scores, generated hypotheses, the host mirror and the device-built tangent mask."""
    S, tmk = small_case["scene"]["surfels"], small_case["scene"]["tangent_mask"]
    gm, ref = gpu_small["model"], gpu_small["scene"]
    d_s, d_t = gpu_ctx.alloc(S.nbytes), gpu_ctx.alloc(tmk.nbytes)
    gpu_ctx.h2d(d_s, S)
    gpu_ctx.h2d(d_t, np.ascontiguousarray(tmk, np.int32))
    gpu_ctx.synchronize()
    dev = capi.Scene.from_device(gpu_ctx, d_s, len(S), d_t)
    poses = random_poses_near_truth(small_case, 30, seed=9)
    a, b = dev.score_hypotheses(gm, poses, DT), ref.score_hypotheses(gm, poses, DT)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[0].max() > 100
    cloud, mask = dev.download()
    assert np.array_equal(cloud.view(np.uint32), S.view(np.uint32)) and np.array_equal(mask, tmk)
    ta, tb = dev.tangent_points(), ref.tangent_points()
    assert np.array_equal(ta[0], tb[0]) and np.array_equal(ta[1].view(np.uint32), tb[1].view(np.uint32))
    # without a mask: none yet; the device-side tangent_mask_ follows
    bare = capi.Scene.from_device(gpu_ctx, d_s, len(S))
    twin = capi.Scene(gpu_ctx, S, np.zeros(len(S), np.int32))
    m1, c1, r1 = bare.auto_tangent_mask(30)
    m2, c2, r2 = twin.auto_tangent_mask(30)
    assert np.array_equal(m1, m2) and np.array_equal(c1, c2) and np.array_equal(r1.view(np.uint32), r2.view(np.uint32))
    gpu_ctx.free(d_s)
    gpu_ctx.free(d_t)
