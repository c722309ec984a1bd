#!/usr/bin/env python3
"""Kernel times of the scoring pass for BASELINE configs 3 / 4 (cylinder 500k, plane2 1M), culled vs exhaustive."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from triplet_match_amd import capi, synth  # noqa: E402
from test_gpu_fullsize_traits import _pick_cylinder_samples, _pick_plane_samples  # noqa: E402

ctx = capi.Context(0)


def run(name, K, model, msamples, S, tm, ssamples, threshold):
    ms = np.zeros(capi.STATE_FLOATS, np.float32)
    ms[1] = threshold
    mstate = capi.traits_init_from_samples(K, ms, np.concatenate([model[i] for i in msamples]))
    sstate = capi.traits_init_from_samples(K, ms, np.concatenate([S[i] for i in ssamples]))
    radius = float(mstate[0]) if K == capi.CYLINDER else 0.0
    gs = capi.Scene(ctx, S, tm)
    import time
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
    ctx.enable_timing(True)
    out = {}
    for mode, label in ((capi.SCORE_CULLED, "culled"), (capi.SCORE_BRUTEFORCE, "exhaustive")):
        gp.score_hypotheses(gm, pg, 3.0, traits=K, mode=mode)
        t = []
        for _ in range(3):
            gp.score_hypotheses(gm, pg, 3.0, traits=K, mode=mode)
            t.append(ctx.last_score_timing()["ms"])
        out[label] = float(np.mean(t))
    ctx.enable_timing(False)
    print("%s: %d hypotheses x %d points: project+prepare %.1f ms, scoring culled %.2f ms, exhaustive %.2f ms"
          % (name, len(pg), len(S), t_proj * 1e3, out["culled"], out["exhaustive"]))


rng = np.random.default_rng(2)
model, _ = synth.synth_pipe_section(50_000, seed=3)
sc = synth.synth_pipe_plant(500_000, 50_000, seed=4)
run("config 3 (cylinder, 500k pipe plant)", capi.CYLINDER, model, _pick_cylinder_samples(model, np.ones(len(model), bool), rng),
    sc["surfels"], sc["tangent_mask"], _pick_cylinder_samples(sc["surfels"], sc["pipe_id"] == 0, rng), 0.05)
rng = np.random.default_rng(3)
model, _ = synth.synth_wall_panel(15_000, seed=5)
sc = synth.synth_indoor(1_000_000, 15_000, seed=6)
run("config 4 (plane2, 1M indoor scan)", capi.PLANE2, model, _pick_plane_samples(model, np.ones(len(model), bool), rng),
    sc["surfels"], sc["tangent_mask"], _pick_plane_samples(sc["surfels"], sc["wall_id"] == 0, rng), 0.05)
