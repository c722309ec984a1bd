#!/usr/bin/env python3
"""Randomised parity sweep: many small seeded cases with varying sizes and parameters, every stage of the
path compared bit for bit with the oracle (scores to 1e-9). usage: tools/parity_sweep.py [n_cases] [seed0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tmo  # noqa: E402
from triplet_match_amd import capi, synth  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def one_case(ctx, seed):
    rng = np.random.default_rng(seed)
    n_model = int(rng.integers(300, 4000))
    copies = int(rng.integers(1, 4))
    n_scene = int(max(n_model * copies * 1.3, rng.integers(2000, 30000)))
    scale = tuple(float(x) for x in (0.6 + 0.8 * rng.random(3)))
    model, subset = synth.synth_model(n_model, seed=seed, scale=scale)
    res = synth.resolution(model)
    scene = synth.synth_scene(n_scene, n_model, seed=seed + 1, copies=copies, model_resolution=res,
                              upright=bool(rng.integers(0, 2)), noise_sigma_res=float(rng.choice([0.0, 0.1, 0.3])))
    S, tm = scene["surfels"].copy(), scene["tangent_mask"]
    if rng.random() < 0.3:  # a few non-finite points
        S[rng.integers(0, n_scene, 5), rng.integers(0, 3, 5)] = np.nan
    dsc = float(rng.choice([10.0, 20.0, 33.0]))
    ang = float(np.deg2rad(rng.choice([5.0, 10.0, 15.0])))
    dt = float(rng.choice([1.0, 2.0, 3.0, 4.5]))
    minf, maxf = float(rng.choice([0.1, 0.2, 0.3])), float(rng.choice([0.8, 1.0]))
    force_up = bool(rng.integers(0, 2))
    gm = capi.Model(ctx, model, res, dsc, ang)
    gm.build_voxels()
    om = tmo.Model(model, res, dsc, ang)
    om.build_voxels(threads=8)
    assert np.array_equal(gm.voxels(), om.voxels()), "voxels"
    try:
        gm.build_pairs(subset, minf, maxf)
        ok_pairs = True
    except capi.TmError:
        ok_pairs = False
    rc = om.build_pairs(subset, minf, maxf)
    assert (rc == 0) == ok_pairs, "pair table status"
    if not ok_pairs:
        return "empty table"
    assert gm.pair_counts()["entries"] == om.pair_count(), "pair count"
    gs = capi.Scene(ctx, S, tm)
    os_ = tmo.Scene(S, tm)
    mask = None
    if rng.random() < 0.5:
        mask = (rng.random(n_scene) < 0.1).astype(np.int32)
        gs.set_mask(mask)
        os_.set_mask(mask)
    pairs = synth.sample_scene_pairs(dict(surfels=np.nan_to_num(S), tangent_mask=tm), om.diameter, 12, 4, seed=seed,
                                     min_diameter_factor=minf, max_diameter_factor=maxf)
    if len(pairs) == 0:
        return "no pairs"
    ql = int(rng.choice([200, 50, 7]))
    pg, vg, total = gs.generate_hypotheses(gm, pairs, minf, maxf, force_up=force_up, query_limit=ql)
    po, vo = os_.generate_hypotheses(om, pairs, minf, maxf, force_up=force_up, query_limit=ql)
    assert total == len(po) and np.array_equal(vg, vo) and np.array_equal(bits(pg), bits(po)), "generate"
    if len(po) == 0:
        return "no hypotheses"
    poses = po[:400]
    co, so, _ = os_.score_hypotheses(om, poses, 0.5, dt, threads=8)
    for mode in (capi.SCORE_CULLED, capi.SCORE_BRUTEFORCE):
        cg, sg = gs.score_hypotheses(gm, poses, dt, mode=mode)
        assert np.array_equal(cg, co), "counts mode %d" % mode
        assert np.allclose(sg, so, rtol=1e-9, atol=1e-15), "scores"
    sub = np.sort(rng.choice(n_scene, size=n_scene // 3, replace=False)).astype(np.int32)
    cs, _ = gs.score_hypotheses(gm, poses[:100], dt, subset=sub)
    cso, _, _ = os_.score_hypotheses(om, poses[:100], 0.5, dt, subset=sub, threads=8)
    assert np.array_equal(cs, cso), "subset counts"
    centre = int(pairs[0, 0])
    nn = gs.radius_subset(centre, om.diameter)
    assert np.array_equal(nn, os_.radius_subset(centre, om.diameter)), "radius subset"
    ce, se, de, ve = gs.score_hypotheses_early_drop(gm, poses[:60], dt, 0.5, subset=nn)
    cb, sb, db, vb = gs.score_hypotheses_early_drop_ball(gm, poses[:60], np.full(60, centre, np.uint32)[:len(poses[:60])],
                                                         om.diameter, dt, 0.5)
    assert np.array_equal(ce, cb) and np.array_equal(se, sb) and np.array_equal(de, db) and np.array_equal(ve, vb), "ed ball"
    for h in range(min(25, len(poses))):
        sc, mc, score, saved = os_.project(om, poses[h], 0.5, dt, subset=nn, early_out=True)
        assert ce[h] == len(sc) and ve[h] == saved, "early drop %d" % h
        assert abs(se[h] - score) <= 1e-9 * abs(score) + 1e-12 * max(1, len(sc)), \
            "early drop score %r vs %r (%d corrs, dropped %d)" % (se[h], score, len(sc), de[h])
    b = int(np.argmax(co))
    scg, mcg, sgx = gs.extract_correspondences(gm, poses[b], dt)
    sco, mco, sox, _ = os_.project(om, poses[b], 0.5, dt)
    assert np.array_equal(scg, sco) and np.array_equal(mcg, mco), "extract"
    return "ok (%d model, %d scene, %d hyps, best %d)" % (n_model, n_scene, len(po), int(co.max()))


def traits_case(ctx, seed):
    """projector metrics: cylinder on a small pipe plant / plane2 on a small room, device-resident projection,
    generation and scoring (culled and exhaustive) against the oracle"""
    rng = np.random.default_rng(seed)
    if seed % 2 == 0:
        K = capi.CYLINDER
        n_model = int(rng.integers(1500, 5000))
        model, _ = synth.synth_pipe_section(n_model, seed=seed)
        sc = synth.synth_pipe_plant(int(rng.integers(20000, 50000)), n_model, seed=seed + 1,
                                    noise_sigma=float(rng.choice([0.0, 0.002])))
        S, tm = sc["surfels"], sc["tangent_mask"]
        on = np.nonzero(sc["pipe_id"] == int(rng.integers(0, 3)))[0]
        msel = [0, int(np.argmin(np.abs(model[:, 4:7] @ model[0, 4:7])))]
        a = int(on[rng.integers(len(on))])
        ssel = [a, int(on[np.argmin(np.abs(S[on, 4:7] @ S[a, 4:7]))])]
        thr = 0.05
    else:
        K = capi.PLANE2
        n_model = int(rng.integers(800, 3000))
        model, _ = synth.synth_wall_panel(n_model, seed=seed)
        sc = synth.synth_indoor(int(rng.integers(30000, 80000)), n_model, seed=seed + 1)
        S, tm = sc["surfels"], sc["tangent_mask"]
        on = np.nonzero(sc["wall_id"] == int(rng.integers(0, 3)))[0]
        msel = [0, int(np.argmax(np.linalg.norm(model[:, 0:3] - model[0, 0:3], axis=1))), n_model // 2]
        ssel = [int(on[0]), int(on[np.argmax(np.linalg.norm(S[on, 0:3] - S[on[0], 0:3], axis=1))]), int(on[len(on) // 2])]
        thr = 0.05
    ms = np.zeros(capi.STATE_FLOATS, np.float32)
    ms[1] = thr
    mstate = capi.traits_init_from_samples(K, ms, np.concatenate([model[i] for i in msel]))
    sstate = capi.traits_init_from_samples(K, ms, np.concatenate([S[i] for i in ssel]))
    if mstate is None or sstate is None:
        return "samples rejected"
    radius = float(mstate[0]) if K == capi.CYLINDER else 0.0
    gs = capi.Scene(ctx, S, tm)
    gp = gs.project(K, sstate)
    su, tmask = gp.download()
    smp = rng.integers(0, len(S), 300)
    for i in smp:
        uvw = tmo.traits_project(K, sstate, S[i, 0:3])
        assert (uvw is None) == bool(np.isnan(su[i, 0])), "project validity"
        if uvw is not None:
            assert np.array_equal(bits(su[i, 0:3]), bits(uvw)), "project"
    mu, mv = ctx.traits_project_cloud(K, mstate, model)
    mu = mu[mv == 1]
    if len(mu) < 100:
        return "model projected away"
    res = synth.resolution(mu)
    subset = np.nonzero(np.linalg.norm(mu[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
    gm = capi.Model(ctx, mu, res)
    gm.set_metric(K, radius)
    gm.build_voxels()
    om = tmo.Model(mu, res)
    om.set_metric(K, radius)
    om.build_voxels(threads=8)
    assert np.array_equal(gm.voxels(), om.voxels()), "voxels"
    try:
        gm.build_pairs(subset)
    except capi.TmError:
        assert om.build_pairs(subset) != 0
        return "empty table"
    assert om.build_pairs(subset) == 0
    os_ = tmo.Scene(su, tmask)
    near = tmask * (np.abs(np.nan_to_num(su[:, 2], nan=1e9)) < (thr / radius if radius else thr))
    pairs = synth.sample_scene_pairs(dict(surfels=np.nan_to_num(su), tangent_mask=near), gm.diameter, 20, 3, seed=seed)
    if len(pairs) == 0:
        return "no pairs"
    pg, vg, total = gp.generate_hypotheses(gm, pairs)
    po, vo = os_.generate_hypotheses(om, pairs)
    assert total == len(po) and np.array_equal(vg, vo) and np.array_equal(bits(pg), bits(po)), "generate"
    if len(po) == 0:
        return "no hypotheses"
    poses = po[:300]
    co, so, _ = os_.score_hypotheses(om, poses, 0.5, 3.0, threads=8)
    for mode in (capi.SCORE_CULLED, capi.SCORE_BRUTEFORCE):
        cg, sg = gp.score_hypotheses(gm, poses, 3.0, traits=K, mode=mode)
        assert np.array_equal(cg, co), "traits counts mode %d" % mode
        assert np.allclose(sg, so, rtol=1e-9, atol=1e-12), "traits scores"
    return "ok traits %d (%d hyps, best %d)" % (K, len(po), int(co.max()))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ctx = capi.Context(0)
    bad = 0
    for k in range(n):
        try:
            msg = traits_case(ctx, seed0 + 7 * k) if os.environ.get("TM_SWEEP_TRAITS") else one_case(ctx, seed0 + 7 * k)
        except AssertionError as e:
            bad += 1
            msg = "MISMATCH: %s" % e
        print("case %3d seed %5d: %s" % (k, seed0 + 7 * k, msg), flush=True)
    print("%d cases, %d mismatches" % (n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
