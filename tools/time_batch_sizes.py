#!/usr/bin/env python3
"""Full-scene scoring rate against the batch size on the configs[1] clouds (100k model, 1M scene), device-resident:
hypotheses/s for 1k .. 1M hypotheses per call (HIP events around tm_score_hypotheses_dev)."""
import os
import sys

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
H = 1 << 20
pairs = synth.sample_scene_pairs(scene, gm.diameter, max(64, H // 400), 24, seed=13)
d_ij = ctx.alloc(pairs.nbytes)
ctx.h2d(d_ij, pairs)
d_poses, d_prov = ctx.alloc(H * 64), ctx.alloc(H * 16)
n_hyp, total = gs.generate_hypotheses_dev(gm, d_ij, len(pairs), d_poses, d_prov, H)
print("%d hypotheses generated (of %d)" % (n_hyp, total))
d_counts, d_scores = ctx.alloc(H * 4), ctx.alloc(H * 8)
for n in (1024, 4096, 16384, 65536, 262144, n_hyp):
    n = min(n, n_hyp)
    best = 1e30
    for rep in range(4):
        ctx.synchronize()
        ctx.time_begin()
        gs.score_hypotheses_dev(gm, d_poses, n, 3.0, d_counts, d_scores, mode=capi.SCORE_CULLED)
        best = min(best, ctx.time_end())
    print("%8d hypotheses: %8.3f ms  %6.2f M hypotheses/s" % (n, best, n / best / 1e3))
