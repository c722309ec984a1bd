#!/usr/bin/env python3
"""Times the pieces of model<Point>::init at the bench size (100k points)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triplet_match_amd import capi, synth  # noqa: E402

model, subset = synth.synth_model(100_000, seed=1)
ctx = capi.Context(0)
capi.Model(ctx, model[:2000], 0.02).close()
for rep in range(3):
    t0 = time.perf_counter()
    res, _ = ctx.cloud_resolution(model)
    t1 = time.perf_counter()
    gm = capi.Model(ctx, model, res)
    t2 = time.perf_counter()
    gm.build_voxels()
    t3 = time.perf_counter()
    gm.build_pairs(subset)
    t4 = time.perf_counter()
    print("resolution %.1f ms, create %.1f ms, voxels %.1f ms, pairs %.1f ms" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
    gm.close()
