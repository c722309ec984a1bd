#!/usr/bin/env python3
"""Scoring-pass figures for BASELINE configs 3 / 4 / 5 (cylinder 500k, plane2 1M, 32 models x 4M) on one GPU:
kernel time grouped / culled vs exhaustive, hypotheses/s, Gpoints/s, the BASELINE formula (40E + 32P + 64H) next
to the bytes the evaluations really request (physical model, as bench.py), and the oracle's project_ on a
sample of the same hypotheses on the host cores. `frac` = measured L2->fabric bytes / kernel time / 8 TB/s (the one
basis of bench.py; null without --traffic); the formula and the requested-bytes model are side fields only.
Prints one JSON object per config.
  --only 3|4|5.0|5.15|5.31   one configuration only (tools/pmc_configs.sh profiles them one at a time)
  --traffic FILE             L2->fabric bytes per launch of the grouped kernel from those PMC passes
                             ({config key: bytes}): adds measured_fabric_GB and frac_measured_hbm"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from triplet_match_amd import capi, synth  # noqa: E402
from test_gpu_fullsize_traits import _pick_cylinder_samples, _pick_plane_samples  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default=None)
ap.add_argument("--traffic", default=None)
ARGS = ap.parse_args()
TRAFFIC = json.load(open(ARGS.traffic)) if ARGS.traffic else {}
ctx = capi.Context(0)


def run(name, K, model, msamples, S, tm, ssamples, threshold, key):
    ms = np.zeros(capi.STATE_FLOATS, np.float32)
    ms[1] = threshold
    mstate = capi.traits_init_from_samples(K, ms, np.concatenate([model[i] for i in msamples]))
    sstate = capi.traits_init_from_samples(K, ms, np.concatenate([S[i] for i in ssamples]))
    radius = float(mstate[0]) if K == capi.CYLINDER else 0.0
    gs = capi.Scene(ctx, S, tm)
    t0 = time.perf_counter()
    gp = gs.project(K, sstate)
    t_proj = time.perf_counter() - t0
    su, tmask = gp.download()
    mu, mv = ctx.traits_project_cloud(K, mstate, model)
    mu = mu[mv == 1]
    res = synth.resolution(mu)
    subset = np.nonzero(np.linalg.norm(mu[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
    gm = capi.Model(ctx, mu, res)
    gm.set_metric(K, radius)
    gm.build_voxels()
    gm.build_pairs(subset)
    near = tmask * (np.abs(np.nan_to_num(su[:, 2], nan=1e9)) < (threshold / radius if radius else threshold))
    pairs = synth.sample_scene_pairs(dict(surfels=su, tangent_mask=near), gm.diameter, 400, 3, seed=13)
    pg, _, _ = gp.generate_hypotheses(gm, pairs, cap=65536)
    report(name, gp, gm, pg, K, su, tmask, mu, res, radius, {"project_and_prepare_ms": round(t_proj * 1e3, 2)}, key=key)


def report(name, gscene, gm, poses, K, S, tmask, M, res, radius, extra, key=None):
    from oracle import tmo
    ctx.enable_timing(True)
    out = {}
    stats = None
    for mode, label in ((capi.SCORE_CULLED, "culled"), (capi.SCORE_BRUTEFORCE, "exhaustive")):
        ctx.enable_timing(1)  # a window with the stage counters: one launch of the batch
        gscene.score_hypotheses(gm, poses, 3.0, traits=K, mode=mode)
        cnt = ctx.score_timing_totals()
        ctx.enable_timing(2)  # the timed launches: events only, the kernel every caller outside a window runs (bench.py)
        for _ in range(3):
            counts, _ = gscene.score_hypotheses(gm, poses, 3.0, traits=K, mode=mode)
        tot = ctx.score_timing_totals()
        out[label] = tot["ms"] / tot["launches"]
        if mode == capi.SCORE_CULLED:
            stats = {k: v / cnt["launches"] for k, v in cnt.items() if k not in ("launches", "ms")}
    ctx.enable_timing(False)
    H, E, P = len(poses), stats["evaluations"], stats["passing"]
    alg = 40.0 * E + 32.0 * P + 64.0 * H
    scene_bytes = (6144.0 + 1408.0) * stats["staged_chunks"] if stats["staged_chunks"] else 12.0 * E
    phys = scene_bytes + stats["in_grid"] + 16.0 * stats["near"] + 32.0 * P + 64.0 * H
    frac_formula = alg / (out["culled"] * 1e-3) / 8e12
    frac_physical = phys / (out["culled"] * 1e-3) / 8e12
    # CPU beside it: the oracle's project_ on a sample, hardware_concurrency() - 1 threads
    threads = max(1, (os.cpu_count() or 2) - 1)
    om = tmo.Model(M, res)
    if K != capi.IDENTITY:
        om.set_metric(K, radius)
    om.set_voxels(gm.voxels())
    os_ = tmo.Scene(S, tmask)
    k = min(H, 4 * threads)
    sel = np.linspace(0, H - 1, k).astype(np.int64)
    t0 = time.perf_counter()
    co, _, ev = os_.score_hypotheses(om, poses[sel], 0.5, 3.0, threads=threads)
    dt = time.perf_counter() - t0
    measured = TRAFFIC.get(key) if key else None
    if measured:
        extra = dict(extra, measured_fabric_GB=round(measured / 1e9, 2),
                     frac_measured_hbm=round(measured / (out["culled"] * 1e-3) / 8e12, 4))
    # one basis, as bench.py: measured L2->fabric bytes / kernel time / 8 TB/s (Infinity Cache + HBM: an upper bound
    # of the HBM traffic); null without a --traffic file. The formula and the requested-bytes model are side fields.
    frac = extra.get("frac_measured_hbm") if measured else None
    basis = ("measured_l2_fabric_bytes: Infinity Cache + HBM, upper bound of HBM" if measured else
             "unavailable: no PMC pass for this configuration (tools/pmc_configs.sh)")
    if measured:
        extra = dict(extra, infinity_cache_frac=round(measured / (out["culled"] * 1e-3) / 8.6e12, 4))
    rec = {"config": name, "hypotheses": H, "scene_points": int(len(S)), "model_points": int(len(M)),
           "kernel_ms_culled": round(out["culled"], 3), "kernel_ms_exhaustive": round(out["exhaustive"], 3),
           "hypotheses_per_s": round(H / (out["culled"] * 1e-3), 1),
           "gpoints_per_s": round(E / (out["culled"] * 1e-3) / 1e9, 2),
           "evaluations": int(E), "passing": int(P),
           "in_grid": int(stats["in_grid"]), "near": int(stats["near"]), "staged_chunks": int(stats["staged_chunks"]),
           "formula_GBs": round(alg / (out["culled"] * 1e-3) / 1e9, 1),
           "physical_GBs": round(phys / (out["culled"] * 1e-3) / 1e9, 1),
           "frac": None if frac is None else round(frac, 4), "frac_basis": basis,
           "frac_formula_raw": round(frac_formula, 4), "frac_physical": round(frac_physical, 4),
           "cpu_port": {"hypotheses_per_s": round(k / dt, 2), "cores": threads, "sample": int(k),
                        "gpu_counts_equal_oracle_on_sample": bool(np.array_equal(co, counts[sel]))}}
    rec.update(extra)
    print(json.dumps(rec), flush=True)


def want(key):
    return ARGS.only is None or ARGS.only == key


if want("3"):
    rng = np.random.default_rng(2)
    model, _ = synth.synth_pipe_section(50_000, seed=3)
    sc = synth.synth_pipe_plant(500_000, 50_000, seed=4)
    run("config 3 (cylinder, 500k pipe plant)", capi.CYLINDER, model, _pick_cylinder_samples(model, np.ones(len(model), bool), rng),
        sc["surfels"], sc["tangent_mask"], _pick_cylinder_samples(sc["surfels"], sc["pipe_id"] == 0, rng), 0.05, "3")
if want("4"):
    rng = np.random.default_rng(3)
    model, _ = synth.synth_wall_panel(15_000, seed=5)
    sc = synth.synth_indoor(1_000_000, 15_000, seed=6)
    run("config 4 (plane2, 1M indoor scan)", capi.PLANE2, model, _pick_plane_samples(model, np.ones(len(model), bool), rng),
        sc["surfels"], sc["tangent_mask"], _pick_plane_samples(sc["surfels"], sc["wall_id"] == 0, rng), 0.05, "4")

# config 5: 32 models x 4M-point scene, one 64k batch per model (three of them reported)
if ARGS.only is None or ARGS.only.startswith("5"):
    models, scene = synth.synth_multi_model_scene(4_000_000, 32, seed=7)
    S, tm = scene["surfels"], scene["tangent_mask"]
    gs = capi.Scene(ctx, S, tm)
    for k in (0, 15, 31):
        if not want("5.%d" % k):
            continue
        M, sub, res = models[k]
        gm = capi.Model(ctx, M, res)
        gm.build_voxels()
        gm.build_pairs(sub)
        own = dict(surfels=S, tangent_mask=tm * (scene["instance_id"] == k))
        pairs = np.concatenate([synth.sample_scene_pairs(own, gm.diameter, 100, 4, seed=13 + k),
                                synth.sample_scene_pairs(dict(surfels=S, tangent_mask=tm), gm.diameter, 300, 4, seed=113 + k)])
        pairs = np.ascontiguousarray(pairs[np.random.default_rng(k).permutation(len(pairs))])
        pg, _, _ = gs.generate_hypotheses(gm, pairs, cap=65536)
        report("config 5 (model %d of 32, %d points, 4M scene)" % (k, len(M)), gs, gm, pg, capi.IDENTITY, S, tm, M, res, 0.0, {},
               key="5.%d" % k)
        gm.close()
