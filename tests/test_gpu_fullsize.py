"""Full-size parity (BASELINE.json configs[1]: 100k-pt model, 1M-pt scene, 64k hypotheses/batch)
through size-independent properties, all through the C ABI:

  * the voxel NN grid (1.7e8 voxels) equals the oracle's own table (SHA-256 committed under tests/golden/,
    built offline on the CPU), the kd-tree kernel equals the exhaustive kernel, and sampled voxels hold a
    true nearest neighbour (fp64 check against all 100k points);
  * the pair table (1.4e7 entries) equals the oracle's, keys, counts and container order;
  * generated hypotheses for a sample of pairs are bit-identical to the oracle's;
  * culled scoring == exhaustive scoring on the whole 64k batch (counts bit-exact, fixed-point scores equal);
  * additivity: counts over a partition of the scene add up to the full-scene counts;
  * mask idempotence, and the oracle's project_ on a sample of the batch incl. its best hypotheses;
  * the planted poses score what the planted instances hold, and the batch's best is one of them.
"""
import os

import numpy as np
import pytest

from oracle import tmo
from triplet_match_amd import capi, synth

pytestmark = pytest.mark.gpu

DT = 3.0
N_MODEL, N_SCENE, H = 100_000, 1_000_000, 65_536


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def full(gpu_ctx):
    model, subset = synth.synth_model(N_MODEL, seed=1)
    res = synth.resolution(model)
    scene = synth.synth_scene(N_SCENE, N_MODEL, seed=2, copies=4, model_resolution=res)
    gm = capi.Model(gpu_ctx, model, res)
    gm.build_voxels()
    gm.build_pairs(subset)
    gs = capi.Scene(gpu_ctx, scene["surfels"], scene["tangent_mask"])
    pairs = synth.sample_scene_pairs(scene, gm.diameter, max(64, H // 400), 24, seed=13)
    poses, prov, total = gs.generate_hypotheses(gm, pairs, cap=H)
    assert len(poses) == H and total >= H
    counts, scores = gs.score_hypotheses(gm, poses, DT, mode=capi.SCORE_CULLED)
    return dict(model=model, subset=subset, res=res, scene=scene, gm=gm, gs=gs, pairs=pairs,
                poses=poses, prov=prov, counts=counts, scores=scores)


@pytest.fixture(scope="module")
def oracle_full(full):
    """Oracle model with its own pair table; the voxel table is imported from the GPU build, whose
    correctness at this size is what test_voxel_grid_full_size establishes."""
    import hashlib
    import json
    om = tmo.Model(full["model"], full["res"])
    vox = full["gm"].voxels()
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_voxel_grid.json")))
    # ... which the oracle built on its own, offline (make_voxel_sha.py): checked before the table is handed over
    assert hashlib.sha256(np.ascontiguousarray(vox, np.uint32).tobytes()).hexdigest() == gold["sha256_kd_tree"]
    om.set_voxels(vox)
    assert om.build_pairs(full["subset"]) == 0
    os_ = tmo.Scene(full["scene"]["surfels"], full["scene"]["tangent_mask"])
    return dict(model=om, scene=os_)


def test_voxel_grid_full_size(gpu_ctx, full):
    gm = full["gm"]
    vox = gm.voxels()
    ext, to_voxel = gm.grid()
    assert vox.size == int(ext[0]) * int(ext[1]) * int(ext[2]) > 1e8
    assert vox.max() < N_MODEL
    # the ORACLE's own table of this model, built offline on the CPU (tests/golden/make_voxel_sha.py: once with
    # its kd-tree, once exhaustively): the full-size result, ties included, is pinned by an implementation that
    # shares no code with the GPU kernels
    import hashlib
    import json
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_voxel_grid.json")))
    assert gold["extents"] == [int(e) for e in ext] and gold["voxels"] == vox.size
    sha = hashlib.sha256(np.ascontiguousarray(vox, np.uint32).tobytes()).hexdigest()
    assert sha == gold["sha256_kd_tree"]
    if "sha256_exhaustive" in gold:
        assert sha == gold["sha256_exhaustive"]
    # kd-tree kernel == exhaustive kernel (same tie rule), whole table
    with gpu_ctx.option("voxel_bruteforce", 1):
        gb = capi.Model(gpu_ctx, full["model"], full["res"])
        gb.build_voxels()
        vb = gb.voxels()
    assert np.array_equal(vox, vb)
    del vb
    gb.close()
    with gpu_ctx.option("voxel_seed", 0):  # ... the packets without their super-brick's candidate
        gu = capi.Model(gpu_ctx, full["model"], full["res"])
        gu.build_voxels()
        assert np.array_equal(vox, gu.voxels())
        gu.close()
    with gpu_ctx.option("voxel_packet", 0):  # ... and the per-voxel descent == the packet descent (the default)
        gp = capi.Model(gpu_ctx, full["model"], full["res"])
        gp.build_voxels()
        assert np.array_equal(vox, gp.voxels())
        gp.close()
    # sampled voxels: the stored point is a nearest neighbour of the voxel centre (fp64, all points)
    rng = np.random.default_rng(3)
    T = np.asarray(to_voxel, np.float64).reshape(4, 4).T  # column-major -> matrix
    Tinv = np.linalg.inv(T)
    P = full["model"][:, 0:3].astype(np.float64)
    lin = rng.integers(0, vox.size, 1500)
    ex, ey = int(ext[0]), int(ext[1])
    ijk = np.stack([lin % ex, (lin // ex) % ey, lin // (ex * ey)], 1).astype(np.float64)
    centres = (np.c_[ijk, np.ones(len(lin))] @ Tinv.T)[:, :3]
    for c0 in range(0, len(lin), 100):
        c = centres[c0:c0 + 100]
        d = np.linalg.norm(c[:, None, :] - P[None, :, :], axis=2)
        dmin = d.min(1)
        dsel = d[np.arange(len(c)), vox[lin[c0:c0 + 100]]]
        assert np.all(dsel <= dmin * (1 + 1e-5) + 1e-7)


def test_pair_table_full_size(full, oracle_full):
    g, o = full["gm"], oracle_full["model"]
    pc = g.pair_counts()
    assert pc["entries"] == o.pair_count() > 1e7 and pc["valid_pairs"] == o.valid_count()
    bg, bo = g.feature_bounds(), o.feature_bounds()
    assert np.array_equal(bits(bg[0]), bits(bo[0])) and np.array_equal(bits(bg[1]), bits(bo[1]))
    keys, offs, pairs = g.table()
    okeys, ocounts = o.table_keys()
    assert np.array_equal(keys, okeys)
    assert np.array_equal(np.diff(offs.astype(np.int64)), ocounts.astype(np.int64))
    # container order: the first 200 (query_limit) entries of every key, all entries of the 20 smallest keys
    small = set(np.argsort(ocounts)[:20].tolist())
    for k in range(len(keys)):
        cap = int(ocounts[k]) if k in small else 200
        ent, total = o.query_key(keys[k], cap=cap)
        assert total == ocounts[k]
        assert np.array_equal(pairs[offs[k]:offs[k] + len(ent)], ent), keys[k]


def test_generation_full_size(full, oracle_full):
    sel = full["pairs"][:: max(1, len(full["pairs"]) // 150)]
    po, vo = oracle_full["scene"].generate_hypotheses(oracle_full["model"], sel)
    pg, vg, total = full["gs"].generate_hypotheses(full["gm"], sel, cap=len(po) + 8)
    assert total == len(po) == len(pg) > 1000
    assert np.array_equal(vg, vo) and np.array_equal(bits(pg), bits(po))
    # the batch is the prefix of the visit order of all pairs
    prov = full["prov"]
    assert np.all(np.diff(prov[:, 0].astype(np.int64)) >= 0) and prov[:, 3].max() < 200


def test_culled_equals_exhaustive_full_batch(full):
    cb, sb = full["gs"].score_hypotheses(full["gm"], full["poses"], DT, mode=capi.SCORE_BRUTEFORCE)
    assert np.array_equal(cb, full["counts"])
    assert np.array_equal(sb, full["scores"])  # order-free fixed-point sums: identical, not just close
    assert full["counts"].max() > 0.9 * N_MODEL > 10 * np.median(full["counts"])


def test_partition_additivity_and_mask_idempotence(full):
    gs, gm = full["gs"], full["gm"]
    rng = np.random.default_rng(11)
    sel = np.unique(np.r_[rng.integers(0, H, 2000), np.argsort(full["counts"])[-48:]])
    poses = full["poses"][sel]
    part = rng.integers(0, 3, N_SCENE)
    tot_c = np.zeros(len(sel), np.int64)
    tot_s = np.zeros(len(sel), np.float64)
    for k in range(3):
        sub = np.nonzero(part == k)[0].astype(np.int32)
        c, s = gs.score_hypotheses(gm, poses, DT, subset=sub)
        tot_c += c
        tot_s += s
    assert np.array_equal(tot_c, full["counts"][sel].astype(np.int64))
    np.testing.assert_allclose(tot_s, full["scores"][sel], rtol=1e-9, atol=1e-9)
    # mask_ marks points already claimed by a match (nonzero = skipped): an all-zero mask changes
    # nothing; masking one part removes exactly its share, even when a subset names those points
    gs.set_mask(np.zeros(N_SCENE, np.int32))
    c1, s1 = gs.score_hypotheses(gm, poses, DT)
    assert np.array_equal(c1, full["counts"][sel]) and np.array_equal(s1, full["scores"][sel])
    sub0 = np.nonzero(part == 0)[0].astype(np.int32)
    gs.set_mask((part == 0).astype(np.int32))
    c2, _ = gs.score_hypotheses(gm, poses, DT)
    c0, _ = gs.score_hypotheses(gm, poses, DT, subset=sub0)
    gs.set_mask(None)
    c3, _ = gs.score_hypotheses(gm, poses, DT, subset=sub0)
    assert np.array_equal(c2.astype(np.int64) + c3, full["counts"][sel].astype(np.int64))
    assert not c0.any()


def test_oracle_project_on_sample_of_full_batch(full, oracle_full):
    rng = np.random.default_rng(5)
    sel = np.unique(np.r_[rng.integers(0, H, 40), np.argsort(full["counts"])[-24:]])
    threads = max(1, min(64, (os.cpu_count() or 2) - 1))
    co, so, _ = oracle_full["scene"].score_hypotheses(oracle_full["model"], full["poses"][sel], 0.5, DT,
                                                      threads=threads)
    assert np.array_equal(co, full["counts"][sel])
    np.testing.assert_allclose(full["scores"][sel], so, rtol=1e-9, atol=1e-9)
    # correspondences of the best hypothesis (finish_find)
    b = int(np.argmax(full["counts"]))
    sc, mc, score = full["gs"].extract_correspondences(full["gm"], full["poses"][b], DT)
    osc, omc, oscore, _ = oracle_full["scene"].project(oracle_full["model"], full["poses"][b], 0.5, DT)
    assert len(osc) == len(sc) == full["counts"][b]
    order = np.argsort(sc, kind="stable")
    assert np.array_equal(sc[order], osc) and np.array_equal(mc[order], omc)
    assert abs(score - oscore) <= 1e-9 * max(1.0, abs(oscore))


def test_planted_poses_and_best_of_batch(gpu_ctx, full):
    scene = full["scene"]
    truth = np.stack([synth.pose_colmajor(T) for T in scene["poses_scene_to_model"]])
    ct, _ = full["gs"].score_hypotheses(full["gm"], truth, DT)
    # every planted instance holds the model's tangent band; the true pose must see most of it
    n_t = len(full["subset"])
    assert np.all(ct > 0.8 * n_t), (ct, n_t)
    best = int(np.argmax(full["counts"]))
    assert full["counts"][best] > 0.5 * n_t
    # the best hypothesis of the batch aligns one planted instance to within the inlier radius
    Tb = full["poses"][best].reshape(4, 4).T.astype(np.float64)
    S = scene["surfels"]
    errs = []
    for k, Tk in enumerate(scene["poses_scene_to_model"]):
        p = S[scene["instance_id"] == k][:2000, 0:3].astype(np.float64)
        a = p @ Tb[:3, :3].T + Tb[:3, 3]
        b = p @ Tk[:3, :3].T + Tk[:3, 3]
        errs.append(np.sqrt(((a - b) ** 2).sum(1)).mean())
    assert min(errs) < 2 * DT * full["res"], errs


def test_sixteen_million_point_scene(gpu_ctx, full):
    """A scene sixteen times the configs[1] size (16M points, 768 MB of surfels: sixteen copies of every point in
    shuffled order) through the size-independent property that copies offer: every count is exactly sixteen times
    the 1M-point scene's, every fixed-point score sum too (so the normalised scores agree to the last bit), for the
    grouped kernel and the exhaustive one, the ball-restricted pass and the early-drop rule with nothing dropped;
    correspondences of the best pose come back sixteen-fold."""
    S, tmk = full["scene"]["surfels"], full["scene"]["tangent_mask"]
    K = 16
    perm = np.random.default_rng(3).permutation(K * N_SCENE)
    src = (perm % N_SCENE).astype(np.int64)  # big[q] is a copy of point src[q]
    big = np.ascontiguousarray(S[src])
    gs16 = capi.Scene(gpu_ctx, big, np.ascontiguousarray(tmk[src]))
    del big
    gm = full["gm"]
    order = np.argsort(-full["counts"].astype(np.int64), kind="stable")
    sel = np.unique(np.r_[order[:24], np.random.default_rng(4).integers(0, H, 2024)])
    poses = full["poses"][sel]
    c16, s16 = gs16.score_hypotheses(gm, poses, DT, mode=capi.SCORE_CULLED)
    assert np.array_equal(c16, K * full["counts"][sel]) and c16.max() > K * 90_000
    assert np.array_equal(s16, K * full["scores"][sel])  # sums of 2^-40 fixed-point terms: exact multiples
    cb, sb = gs16.score_hypotheses(gm, poses[:160], DT, mode=capi.SCORE_BRUTEFORCE)
    assert np.array_equal(cb, c16[:160]) and np.array_equal(sb, s16[:160])
    # ball-restricted pass and the early-drop rule (accept_prob 0: nothing drops) around a first point of the big scene:
    # centre q of the big scene is a copy of point src[q] of the 1M scene
    prov = full["prov"][sel]
    first_small = full["pairs"][prov[:, 0], 0].astype(np.int64)
    where = np.full(N_SCENE, -1, np.int64)
    where[src[::-1]] = np.arange(K * N_SCENE - 1, -1, -1)  # the first copy of every point
    centers16 = where[first_small].astype(np.uint32)
    few = slice(0, 512)
    cs, ss = full["gs"].score_hypotheses_ball(gm, poses[few], first_small[few].astype(np.uint32), gm.diameter, DT)
    cbig, sbig = gs16.score_hypotheses_ball(gm, poses[few], centers16[few], gm.diameter, DT)
    assert np.array_equal(cbig, K * cs) and np.array_equal(sbig, K * ss) and cbig.max() > K * 50_000
    ce, se, de, _ = gs16.score_hypotheses_early_drop_ball(gm, poses[few], centers16[few], gm.diameter, DT, 0.0)
    assert (de < 0).all() and np.array_equal(ce, cbig) and np.array_equal(se, sbig)
    # finish_find for the best pose: sixteen copies of every correspondence, each with its point's model partner
    sc1, mc1, _ = full["gs"].extract_correspondences(gm, full["poses"][order[0]], DT)
    sc16, mc16, _ = gs16.extract_correspondences(gm, full["poses"][order[0]], DT)
    assert len(sc16) == K * len(sc1) and (np.diff(sc16.astype(np.int64)) > 0).all()
    partner = np.full(N_SCENE, -1, np.int64)
    partner[sc1] = mc1
    assert np.array_equal(partner[src[sc16]], mc16.astype(np.int64))
    gs16.close()


def test_curvature_lists_by_lane_groups_equal_per_thread_lists_full_size(gpu_ctx, full):
    """the curvature pass at the configs[1] scene size: the k-nearest lists built by eight lanes per query (default)
    == one thread per query (curv_groups = 0), bit for bit — for the tangent candidates of the 1M-point scene (a sparse
    query list on the scene's tree), for all points of a 300k-point cloud (the queries are the tree's own points), for
    k = 32 (every list register in use) and k = 1; a sample of rows against the oracle's exhaustive neighbour scan"""
    S, tmk = full["scene"]["surfels"], full["scene"]["tangent_mask"]
    idx = np.nonzero(tmk)[0].astype(np.uint32)
    assert len(idx) > 10_000
    rows = {}
    for groups in (1, 0):
        with gpu_ctx.option("curv_groups", groups):
            rows[groups] = (full["gs"].curvature_subset(idx, 30), gpu_ctx.cloud_curvature(S[:300_000], 30),
                            full["gs"].curvature_subset(idx[:5000], 32), full["gs"].curvature_subset(idx[:5000], 1))
    for a, b in zip(rows[1], rows[0]):
        assert np.array_equal(bits(a), bits(b))
    assert (rows[1][0][:, 4] > 0).any()
    sample = np.random.default_rng(8).choice(20_000, 300, replace=False).astype(np.uint32)
    want = tmo.cloud_curvature(S[:20_000], 30)[sample]
    assert np.array_equal(bits(gpu_ctx.cloud_curvature_subset(S[:20_000], sample, 30)), bits(want))
