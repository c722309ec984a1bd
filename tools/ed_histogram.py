#!/usr/bin/env python3
"""Where do the hypotheses of a batch leave the early-drop rule? Histogram of the checkpoint index (0..17, -1 = the
scan ran to the end) over the bench's batch, scored on the neighbourhood of each hypothesis' first point."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triplet_match_amd import capi, synth  # noqa: E402

model, subset = synth.synth_model(100_000, seed=1)
res = synth.resolution(model)
scene = synth.synth_scene(1_000_000, 100_000, seed=2, copies=4, model_resolution=res)
ctx = capi.Context(0)
gm = capi.Model(ctx, model, res)
gm.build_voxels()
gm.build_pairs(subset)
gs = capi.Scene(ctx, scene["surfels"], scene["tangent_mask"])
pairs = synth.sample_scene_pairs(scene, gm.diameter, 164, 24, seed=13)
poses, prov, _ = gs.generate_hypotheses(gm, pairs, cap=65536)
centers = pairs[prov[:, 0], 0].astype(np.uint32)
cnt, sc, dropped, saved = gs.score_hypotheses_early_drop_ball(gm, poses, centers, gm.diameter, 3.0, 0.5)
h = np.bincount(dropped + 1, minlength=19)
print("hypotheses", len(poses), "scan to the end", int(h[0]))
print("dropped at checkpoint k:", {k: int(v) for k, v in enumerate(h[1:]) if v})
