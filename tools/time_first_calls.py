import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
t0 = time.perf_counter()
from triplet_match_amd import capi, synth
model, subset = synth.synth_model(100_000, seed=1)
res = synth.resolution(model)
scene = synth.synth_scene(1_000_000, 100_000, seed=2, copies=4, model_resolution=res)
t1 = time.perf_counter()
ctx = capi.Context(0)
t2 = time.perf_counter()
print("context (HIP start-up) %.1f ms" % ((t2 - t1) * 1e3))
def T(name, fn):
    a = time.perf_counter(); r = fn(); b = time.perf_counter()
    print("%-40s %.1f ms" % (name, (b - a) * 1e3)); return r
T("resolution of 64 points (first kernels)", lambda: ctx.cloud_resolution(model[:64]))
T("resolution of 64 points again", lambda: ctx.cloud_resolution(model[:64]))
T("resolution of the model", lambda: ctx.cloud_resolution(model))
T("resolution of the model again", lambda: ctx.cloud_resolution(model))
gm = T("model create", lambda: capi.Model(ctx, model, res))
T("voxels", gm.build_voxels)
T("pairs", lambda: gm.build_pairs(subset))
gm2 = T("model create again", lambda: capi.Model(ctx, model, res))
T("voxels again", gm2.build_voxels)
T("pairs again", lambda: gm2.build_pairs(subset))
gs = T("scene create", lambda: capi.Scene(ctx, scene["surfels"], scene["tangent_mask"]))
gs2 = T("scene create again", lambda: capi.Scene(ctx, scene["surfels"], scene["tangent_mask"]))
