"""BASELINE.json configs[2] and [3] at their sizes: cylinder_traits on a 500k-pt pipe-plant scene and
plane2_traits on a 1M-pt indoor scan. The per-projector step runs on the device
(tm_scene_create_projected); checked here:

  * the device-resident uvw scene equals the host-buffer flow (tm_traits_project_cloud + tm_scene_create)
    bit for bit, and a sample of it equals the oracle's scalar traits functions;
  * hypotheses generated on both scenes are identical; inlier counts / scores of the whole batch are
    identical between the two scenes and equal the oracle's project_ on a sample incl. the best ones;
  * the best hypothesis explains the planted section / panel of the projector's pipe / wall.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import tmo
from test_gpu_parity import _project_cloud_oracle, bits
from triplet_match_amd import capi, synth

pytestmark = pytest.mark.gpu

DT = 3.0
# SHA-256 of the projected models and of their voxel tables as the ORACLE alone builds them on the CPU
# (tests/golden/make_voxel_sha_configs.py: oracle traits functions per point, oracle kd-tree voxel build)
PINS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c345_voxel_grids.json")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _pick_cylinder_samples(s, sel, rng):
    """two points of the pipe with clearly different normal directions"""
    idx = np.nonzero(sel)[0]
    a = int(idx[rng.integers(len(idx))])
    for _ in range(1000):
        b = int(idx[rng.integers(len(idx))])
        if abs(float(s[a, 4:7] @ s[b, 4:7])) < 0.3:
            return [a, b]
    raise AssertionError("no sample pair")


def _pick_plane_samples(s, sel, rng):
    idx = np.nonzero(sel)[0]
    for _ in range(1000):
        a, b, c = (int(idx[k]) for k in rng.integers(len(idx), size=3))
        d1, d2 = s[b, 0:3] - s[a, 0:3], s[c, 0:3] - s[a, 0:3]
        if np.linalg.norm(np.cross(d1, d2)) > 0.5 * np.linalg.norm(d1) * np.linalg.norm(d2) > 0.05:
            return [a, b, c]
    raise AssertionError("no sample triple")


def _run_case(gpu_ctx, K, model, msamples, scene, ssamples, threshold, target_sel, n_first, min_best_fraction, pin=None):
    rng = np.random.default_rng(17)
    S, tm = scene["surfels"], scene["tangent_mask"]
    ms = np.zeros(capi.STATE_FLOATS, np.float32)
    ms[1] = threshold
    mstate = capi.traits_init_from_samples(K, ms, np.concatenate([model[i] for i in msamples]))
    sstate = capi.traits_init_from_samples(K, ms, np.concatenate([S[i] for i in ssamples]))
    assert mstate is not None and sstate is not None
    radius = float(mstate[0]) if K == capi.CYLINDER else 0.0
    if K == capi.CYLINDER:
        assert abs(float(sstate[0]) - radius) < 0.02 * radius  # the scene pipe has the model's radius

    # --- per-projector step on the device vs the host-buffer flow vs the oracle
    gs = capi.Scene(gpu_ctx, S, tm)
    gp = gs.project(K, sstate)
    pu, ptm = gp.download()
    su, sv = gpu_ctx.traits_project_cloud(K, sstate, S)
    su = su.copy()
    su[sv == 0, 0] = np.nan
    tmask = ((np.linalg.norm(su[:, 9:12], axis=1) > 0.7) & (sv == 1)).astype(np.int32)
    assert np.array_equal(ptm, tmask)
    assert np.array_equal(bits(pu), bits(su)) or np.array_equal(np.isnan(pu), np.isnan(su)) and np.array_equal(
        bits(np.nan_to_num(pu)), bits(np.nan_to_num(su)))
    smp = rng.integers(0, len(S), 4000)
    so, svo = _project_cloud_oracle(K, sstate, S[smp])
    assert np.array_equal(svo, sv[smp])
    ok = svo == 1
    assert np.array_equal(bits(so[ok][:, 0:3]), bits(su[smp][ok][:, 0:3]))
    tg = ok & (np.linalg.norm(S[smp][:, 9:12], axis=1) > 0)
    assert np.array_equal(bits(so[tg][:, 9:12]), bits(su[smp][tg][:, 9:12]))
    assert np.array_equal(bits(so[tg][:, 4:7]), bits(su[smp][tg][:, 4:7]))
    valid_frac = sv.mean()

    # --- model in its own uvw space
    mu, mv = gpu_ctx.traits_project_cloud(K, mstate, model)
    assert mv.mean() > 0.99
    mu = mu[mv == 1]
    res = synth.resolution(mu)
    subset = np.nonzero(np.linalg.norm(mu[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
    gm = capi.Model(gpu_ctx, mu, res)
    gm.set_metric(K, radius)
    gm.build_voxels()
    gm.build_pairs(subset)
    if pin is not None:
        # the GPU's projected model and its voxel table against what the oracle built on its own, BEFORE the table is
        # handed to the oracle's project_ below: that comparison no longer rests on a table nobody else has built
        assert msamples == pin["msamples"]
        # (the positions: what the voxel grid is built from; tangents / normals are compared on a sample above)
        assert _sha(np.ascontiguousarray(mu[:, 0:3], np.float32)) == pin["positions_sha256"], "projected model differs from the oracle's"
        assert [int(e) for e in gm.grid()[0]] == pin["extents"]
        assert _sha(np.ascontiguousarray(gm.voxels(), np.uint32)) == pin["voxel_sha256"], "voxel table differs from the oracle's"

    # --- hypotheses from tangent pairs of the projected scene: identical on both scene objects
    gs2 = capi.Scene(gpu_ctx, su, tmask)
    # sample pairs among the tangent points on the projector's surface (|w| small); plane2's project() keeps
    # every point of the room, and pairs of far-away points only dilute the batch
    near = tmask * (np.abs(np.nan_to_num(su[:, 2], nan=1e9)) < (threshold / radius if radius else threshold))
    pairs = synth.sample_scene_pairs(dict(surfels=su, tangent_mask=near), gm.diameter, n_first, 3, seed=13)
    assert len(pairs) > 100
    pg, vg, total = gp.generate_hypotheses(gm, pairs, cap=65536)
    pg2, vg2, total2 = gs2.generate_hypotheses(gm, pairs, cap=65536)
    assert total == total2 and np.array_equal(vg, vg2) and np.array_equal(bits(pg), bits(pg2))
    assert len(pg) > 2000
    cg, sg = gp.score_hypotheses(gm, pg, DT, traits=K)
    cg2, sg2 = gs2.score_hypotheses(gm, pg, DT, traits=K)
    assert np.array_equal(cg, cg2) and np.array_equal(sg, sg2)
    # sphere-culled kernel (default) == exhaustive kernel under the traits' metric
    cb, sb = gp.score_hypotheses(gm, pg[:8192], DT, traits=K, mode=capi.SCORE_BRUTEFORCE)
    assert np.array_equal(cb, cg[:8192]) and np.array_equal(sb, sg[:8192])

    # --- oracle: generation for a sample of pairs, project_ for a sample of hypotheses incl. the best
    om = tmo.Model(mu, res)
    om.set_metric(K, radius)
    om.set_voxels(gm.voxels())
    assert om.build_pairs(subset) == 0
    os_ = tmo.Scene(su, tmask)
    po, vo = os_.generate_hypotheses(om, pairs[:40])
    pgs, vgs, _ = gp.generate_hypotheses(gm, pairs[:40])
    assert np.array_equal(vgs, vo) and np.array_equal(bits(pgs), bits(po))
    sel = np.unique(np.r_[rng.integers(0, len(pg), 24), np.argsort(cg)[-12:]])
    co, so_, _ = os_.score_hypotheses(om, pg[sel], 0.5, DT, threads=32)
    assert np.array_equal(co, cg[sel])
    np.testing.assert_allclose(sg[sel], so_, rtol=1e-9, atol=1e-12)

    # --- the planted pattern is found. Plain (tangent-free) surface matches the model's plain points at
    # any in-surface pose, so the inlier count alone does not locate the pattern; the reference's
    # arbitration also demands tangent inliers (min_points, include/impl/scene.hpp:114,227): score the
    # tangent points alone (subset pass) and take the sufficient hypothesis with most of them.
    n_target = int(target_sel.sum())
    n_target_t = int((target_sel & (tmask == 1)).sum())
    ct, _ = gp.score_hypotheses(gm, pg, DT, subset=np.nonzero(tmask)[0].astype(np.int32), traits=K)
    best = int(np.argmax(np.where(cg >= min_best_fraction * n_target, ct, 0)))
    best_t = int(ct[best])
    sc, mc, _ = gp.extract_correspondences(gm, pg[best], DT)
    assert int(tmask[sc].sum()) == best_t  # the subset pass counts exactly the tangent correspondences
    on_target = float(target_sel[sc].mean())
    print("traits case: kind %d valid %.3f hyps %d best %d of %d, tangent inliers %d of %d, on-target %.2f"
          % (K, valid_frac, len(pg), cg[best], n_target, best_t, n_target_t, on_target))
    assert cg[best] >= min_best_fraction * n_target, (cg[best], n_target, valid_frac)
    assert best_t >= 0.5 * n_target_t, (best_t, n_target_t)
    assert on_target > 0.8, on_target
    return dict(valid_frac=float(valid_frac), n_hyp=len(pg), best=int(cg[best]), n_target=n_target)


def test_cylinder_traits_pipe_plant_500k(gpu_ctx):
    model, _ = synth.synth_pipe_section(50_000, seed=3)
    sc = synth.synth_pipe_plant(500_000, 50_000, seed=4)
    rng = np.random.default_rng(2)
    msamples = _pick_cylinder_samples(model, np.ones(len(model), bool), rng)
    for pipe, _, _ in sc["instances"][:2]:
        ssamples = _pick_cylinder_samples(sc["surfels"], sc["pipe_id"] == pipe, rng)
        inst = [k for k, it in enumerate(sc["instances"]) if it[0] == pipe][0]
        r = _run_case(gpu_ctx, capi.CYLINDER, model, msamples, sc, ssamples, 0.05, sc["instance_id"] == inst,
                      n_first=400, min_best_fraction=0.5, pin=PINS["c3_pipe_section_cylinder"])
        assert 0.1 < r["valid_frac"] < 0.6  # only this pipe (and what touches its shell) projects


def test_plane2_traits_indoor_1m(gpu_ctx):
    model, _ = synth.synth_wall_panel(15_000, seed=5)
    sc = synth.synth_indoor(1_000_000, 15_000, seed=6)
    rng = np.random.default_rng(3)
    msamples = _pick_plane_samples(model, np.ones(len(model), bool), rng)
    wall = sc["instances"][0][0]
    ssamples = _pick_plane_samples(sc["surfels"], sc["wall_id"] == wall, rng)
    r = _run_case(gpu_ctx, capi.PLANE2, model, msamples, sc, ssamples, 0.05, sc["instance_id"] == 0,
                  n_first=400, min_best_fraction=0.5, pin=PINS["c4_wall_panel_plane2"])
    assert r["valid_frac"] == 1.0  # plane2 project() never rejects (include/impl/plane2_traits.hpp:86-88)
