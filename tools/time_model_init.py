#!/usr/bin/env python3
"""Times the pieces of model<Point>::init at the bench size (100k points): resolution, the curvature of the tangent
candidates (what the subset filter reads), create, voxel grid, pair table."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triplet_match_amd import capi, synth  # noqa: E402

model, subset = synth.synth_model(100_000, seed=1)
cand = np.nonzero(np.linalg.norm(model[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
ctx = capi.Context(0)
capi.Model(ctx, model[:2000], 0.02).close()
for rep in range(3):
    t0 = time.perf_counter()
    res, _ = ctx.cloud_resolution(model)
    t1 = time.perf_counter()
    ctx.cloud_curvature_subset(model, cand, 30)
    t1b = time.perf_counter()
    gm = capi.Model(ctx, model, res)
    t2 = time.perf_counter()
    gm.build_voxels()
    t3 = time.perf_counter()
    gm.build_pairs(subset)
    t4 = time.perf_counter()
    print("resolution %.1f ms, curvature of %d candidates %.1f ms, create %.1f ms, voxels %.1f ms, pairs %.1f ms: %.1f ms in all" %
          ((t1 - t0) * 1e3, len(cand), (t1b - t1) * 1e3, (t2 - t1b) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3))
    gm.close()
