#!/usr/bin/env python3
"""Ball-restricted scoring (tm_score_hypotheses_ball) on the C2 clouds for several batch sizes: grouped kernel
against the round-1 kernel (TM_BALL_UNGROUPED=1), ms per call incl. the host copies."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triplet_match_amd import capi, synth  # noqa: E402

model, subset = synth.synth_model(100_000, seed=1)
res = synth.resolution(model)
scene = synth.synth_scene(1_000_000, 100_000, seed=2, copies=4, model_resolution=res)
ctx = capi.Context(0)
gm = capi.Model(ctx, model, res)
gm.build_voxels()
gm.build_pairs(subset)
gs = capi.Scene(ctx, scene["surfels"], scene["tangent_mask"])
pairs = synth.sample_scene_pairs(scene, gm.diameter, 400, 24, seed=13)
poses, prov, _ = gs.generate_hypotheses(gm, pairs, cap=65536)
centers = pairs[prov[:, 0], 0].astype(np.uint32)
print("hypotheses", len(poses), "distinct centres", len(np.unique(centers)))
for n in (1024, 2048, 4096, 16384, 65536):
    n = min(n, len(poses))
    for env in (None, "ball_ungrouped"):
        if env:
            ctx.set_option(env, 1)
        gs.score_hypotheses_ball(gm, poses[:n], centers[:n], gm.diameter, 3.0)
        t0 = time.perf_counter()
        for _ in range(3):
            c, s = gs.score_hypotheses_ball(gm, poses[:n], centers[:n], gm.diameter, 3.0)
        dt = (time.perf_counter() - t0) / 3
        if env:
            ctx.set_option(env, 0)
        print("n %6d %-18s %.2f ms per call (%.0f hyp/s), max count %d" % (n, env or "grouped", dt * 1e3, n / dt, c.max()), flush=True)
