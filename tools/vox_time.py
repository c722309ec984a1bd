import time, sys
sys.path.insert(0, '.')
import numpy as np
from triplet_match_amd import capi, synth
ctx = capi.Context(0)
model, subset = synth.synth_model(100000, seed=1)
res = synth.resolution(model)
for trial in range(3):
    gm = capi.Model(ctx, model, res)
    ts = []
    for r in range(4):
        t0 = time.perf_counter(); gm.build_voxels(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print("model", trial, "build_voxels ms:", ts, flush=True)
    if trial == 1:
        big = [capi.Model(ctx, model, res) for _ in range(2)]
        for b in big: b.build_voxels()
        del big
