#!/usr/bin/env python3
"""Device memory over repeated model / scene creation, curvature, scoring and search calls: the handles' destroy
calls give everything back (usage after warm-up must not grow). usage (GPU box, repo root): tools/leak_check.py"""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from triplet_match_amd import capi, synth
model, subset = synth.synth_model(5000, seed=1)
res = synth.resolution(model)
sc = synth.synth_scene(200000, 5000, seed=2, copies=3, model_resolution=res)
S, tm = sc["surfels"], sc["tangent_mask"]
ctx = capi.Context(0)
def used():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 1e6
base = None
N_IT = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for it in range(N_IT):
    gm = capi.Model(ctx, model, res); gm.build_voxels(); gm.build_pairs(subset)
    gs = capi.Scene(ctx, S, np.zeros(len(S), np.int32))
    idx = np.nonzero(tm)[0].astype(np.uint32)
    gs.curvature_subset(idx, 30)
    gs.set_tangent_mask(tm)
    pairs = synth.sample_scene_pairs(sc, gm.diameter, 20, 4, seed=it)
    poses, prov, _ = gs.generate_hypotheses(gm, pairs)
    if len(poses):
        centers = pairs[prov[:, 0], 0].astype(np.uint32)
        gs.score_hypotheses(gm, poses, 3.0)
        gs.score_hypotheses_early_drop_ball(gm, poses, centers, gm.diameter, 3.0, 0.5)
        gs.search_batch(gm, pairs, gm.diameter, 3.0, 0.5, True, 1e9, min_diameter_factor=0.2)
    gs.close(); gm.close()
    u = used()
    if it == 4: base = u
    cached = ctx.pool_stats()[0] / 1e6  # released blocks the context keeps for reuse (option pool_mib) are not a leak
    if it == 4: base_live = u - cached
    if it == 4 or it % 25 == 4 or it == N_IT - 1: print("iteration %d: %.0f MB in use, %.0f MB of it in the context's block cache" % (it, u, cached))
print("growth after warm-up: %.1f MB (outside the block cache: %.1f MB); cache reused %d requests, hipMalloc served %d"
      % (used() - base, used() - ctx.pool_stats()[0] / 1e6 - base_live, ctx.pool_stats()[1], ctx.pool_stats()[2]))
