#!/usr/bin/env python3
"""Times tm_scene_create / tm_scene_set_mask (device-side scene preparation) at the bench sizes."""
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
capi.Scene(ctx, scene["surfels"][:1000], scene["tangent_mask"][:1000])  # warm-up (module load)
for rep in range(3):
    t0 = time.perf_counter()
    gs = capi.Scene(ctx, scene["surfels"], scene["tangent_mask"])
    t1 = time.perf_counter()
    mask = (scene["instance_id"] == 0).astype(np.int32)
    gs.set_mask(mask)
    t2 = time.perf_counter()
    gs.set_mask(None)
    t3 = time.perf_counter()
    print("scene_create %.1f ms, set_mask %.1f ms, clear_mask %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
