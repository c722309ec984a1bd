#!/usr/bin/env python3
"""plane2 (config 4: 1M indoor scan in a wall's uvw space): ball-restricted scoring and tangent-list scoring per
call for small and large batches — what a chunk / a whole round of the projected search costs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from triplet_match_amd import capi, synth  # noqa: E402
from test_gpu_fullsize_traits import _pick_plane_samples  # noqa: E402

ctx = capi.Context(0)
K = capi.PLANE2
rng = np.random.default_rng(3)
model, _ = synth.synth_wall_panel(15_000, seed=5)
sc = synth.synth_indoor(1_000_000, 15_000, seed=6)
S, tm = sc["surfels"], sc["tangent_mask"]
ms = np.zeros(capi.STATE_FLOATS, np.float32)
ms[1] = 0.05
mstate = capi.traits_init_from_samples(K, ms, np.concatenate([model[i] for i in _pick_plane_samples(model, np.ones(len(model), bool), rng)]))
sstate = capi.traits_init_from_samples(K, ms, np.concatenate([S[i] for i in _pick_plane_samples(S, sc["wall_id"] == 0, rng)]))
gs = capi.Scene(ctx, S, tm)
gp = gs.project(K, sstate)
su, tmask = gp.download()
mu, mv = ctx.traits_project_cloud(K, mstate, model)
mu = mu[mv == 1]
res = synth.resolution(mu)
subset = np.nonzero(np.linalg.norm(mu[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
gm = capi.Model(ctx, mu, res)
gm.set_metric(K, 0.0)
gm.build_voxels()
gm.build_pairs(subset)
near = tmask * (np.abs(np.nan_to_num(su[:, 2], nan=1e9)) < 0.05)
pairs = synth.sample_scene_pairs(dict(surfels=su, tangent_mask=near), gm.diameter, 100, 3, seed=13)
poses, prov, _ = gp.generate_hypotheses(gm, pairs, cap=65536)
centers = pairs[prov[:, 0], 0].astype(np.uint32)
tangent = np.nonzero(near)[0].astype(np.int32)
print("hypotheses", len(poses), "centres", len(np.unique(centers)), "tangent points", len(tangent), flush=True)
for n in (1600, 16000):
    n = min(n, len(poses))
    for env in (None, "ball_ungrouped"):
        if env:
            ctx.set_option(env, 1)
        gp.score_hypotheses_ball(gm, poses[:n], centers[:n], gm.diameter, 3.0)
        t0 = time.perf_counter()
        for _ in range(3):
            c, s = gp.score_hypotheses_ball(gm, poses[:n], centers[:n], gm.diameter, 3.0)
        dt = (time.perf_counter() - t0) / 3
        if env:
            ctx.set_option(env, 0)
        print("ball   n %6d %-18s %.2f ms per call" % (n, env or "grouped", dt * 1e3), flush=True)
for n in (800, 8000):
    n = min(n, len(poses))
    gp.score_hypotheses_on_points(gm, poses[:n], tangent, 3.0)
    t0 = time.perf_counter()
    for _ in range(3):
        gp.score_hypotheses_on_points(gm, poses[:n], tangent, 3.0)
    print("points n %6d x %d points: %.2f ms per call" % (n, len(tangent), (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    t0 = time.perf_counter()
    for _ in range(3):
        gp.score_hypotheses(gm, poses[:n], 3.0, subset=tangent, traits=K)
    print("subset n %6d (whole-scene kernel): %.2f ms per call" % (n, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
