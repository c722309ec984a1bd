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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ctx = capi.Context(0)
    bad = 0
    for k in range(n):
        try:
            msg = one_case(ctx, seed0 + 7 * k)
        except AssertionError as e:
            bad += 1
            msg = "MISMATCH: %s" % e
        print("case %3d seed %5d: %s" % (k, seed0 + 7 * k, msg), flush=True)
    print("%d cases, %d mismatches" % (n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
