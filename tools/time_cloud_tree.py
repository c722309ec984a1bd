#!/usr/bin/env python3
"""First neighbour query on a 1M-point cloud (pointcloud::knn_inclusive's tree, tm_cloud_create + tm_cloud_knn)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triplet_match_amd import capi, synth  # noqa: E402

model, _ = synth.synth_model(100_000, seed=1)
scene = synth.synth_scene(1_000_000, 100_000, seed=2, copies=4, model_resolution=synth.resolution(model))
S = scene["surfels"]
ctx = capi.Context(0)
capi.Cloud(ctx, S[:1000]).knn(S[:4, :3].copy(), 8)
for rep in range(3):
    t0 = time.perf_counter()
    gc = capi.Cloud(ctx, S)
    t1 = time.perf_counter()
    gc.knn(S[:1000, :3].copy(), 30)
    t2 = time.perf_counter()
    print("tree over %d points %.1f ms, 1000 x 30-NN %.2f ms" % (len(S), (t1 - t0) * 1e3, (t2 - t1) * 1e3))
