#!/usr/bin/env python3
"""Times the one-off pieces of the first find_all_parallel call at the C2 size: model side (resolution, curvature,
create, voxel grid, pair table) and scene side (curvature, create, set_mask)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triplet_match_amd import capi, synth  # noqa: E402

model, subset = synth.synth_model(100_000, seed=1)
res0 = synth.resolution(model)
scene = synth.synth_scene(1_000_000, 100_000, seed=2, copies=4, model_resolution=res0)
S, tm = scene["surfels"], scene["tangent_mask"]
ctx = capi.Context(0)
capi.Model(ctx, model[:2000], 0.02).close()
capi.Scene(ctx, S[:1000], tm[:1000])
ctx.cloud_curvature(model[:2000], 30)
for rep in range(3):
    t = [time.perf_counter()]
    res, _ = ctx.cloud_resolution(model); t.append(time.perf_counter())
    ctx.cloud_curvature(model, 30); t.append(time.perf_counter())
    gm = capi.Model(ctx, model, res); t.append(time.perf_counter())
    gm.build_voxels(); t.append(time.perf_counter())
    gm.build_pairs(subset); t.append(time.perf_counter())
    ctx.cloud_curvature(S, 30); t.append(time.perf_counter())
    gs = capi.Scene(ctx, S, tm); t.append(time.perf_counter())
    gs.set_mask(np.zeros(len(S), np.int32)); t.append(time.perf_counter())
    names = ["model resolution", "model curvature", "model create", "voxels", "pairs", "scene curvature", "scene create",
             "set_mask"]
    print(", ".join("%s %.1f ms" % (n, (b - a) * 1e3) for n, a, b in zip(names, t[:-1], t[1:])), flush=True)
    gm.close()
