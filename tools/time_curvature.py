#!/usr/bin/env python3
"""The k-nearest walks at the configs[1] sizes with eight lanes per query (default) and one thread per query
(curv_groups = 0): curvature of the tangent candidates of the 1M-point scene on the scene's tree, of all points of the
1M-point cloud and the 100k-point model; cloud->resolution(); the neighbour accessor (tm_cloud_knn). TM_DEBUG_TIMES=1 (option debug_times) prints the laps of each call (tree, kernel)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triplet_match_amd import capi, synth  # noqa: E402

model, _ = synth.synth_model(100_000, seed=1)
scene = synth.synth_scene(1_000_000, 100_000, seed=2, copies=4, model_resolution=synth.resolution(model))
S, tmk = scene["surfels"], scene["tangent_mask"]
idx = np.nonzero(tmk)[0].astype(np.uint32)
ctx = capi.Context(0)
gs = capi.Scene(ctx, S, tmk)
gs.curvature_subset(idx[:100], 30)
gc = capi.Cloud(ctx, S)
rng = np.random.default_rng(5)
qs = np.ascontiguousarray(S[rng.choice(len(S), 20000, replace=False), :3] + np.float32(0.001))  # in no spatial order
out = {}
for groups in (1, 0, 1):
    ctx.set_option("curv_groups", groups)
    for name, fn in (("scene subset (%d of %d)" % (len(idx), len(S)), lambda: gs.curvature_subset(idx, 30)),
                     ("cloud, all %d points" % len(S), lambda: ctx.cloud_curvature(S, 30)),
                     ("model, all %d points" % len(model), lambda: ctx.cloud_curvature(model, 30)),
                     ("resolution, %d points" % len(S), lambda: ctx.cloud_resolution(S)[1]),
                     ("resolution, %d points" % len(model), lambda: ctx.cloud_resolution(model)[1]),
                     ("accessor: 30 nearest of 20000 positions", lambda: np.concatenate(gc.knn(qs, 30)[:2], axis=1))):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            r = fn()
            best = min(best, time.perf_counter() - t0)
        out.setdefault(name, {})[groups] = r
        print("curv_groups=%d  %-36s %.2f ms" % (groups, name, best * 1e3), flush=True)
for name, r in out.items():
    assert np.array_equal(r[0].view(np.uint32), r[1].view(np.uint32)), name
print("rows equal")
