"""one-off: kd-tree voxel build == exhaustive kernel at the C2 model size (100k points, 1.7e8 voxels)"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from triplet_match_amd import capi, synth
model, _ = synth.synth_model(100000, seed=1)
res = synth.resolution(model)
ctx = capi.Context(0)
t = time.time(); a = capi.Model(ctx, model, res); a.build_voxels(); ta = time.time() - t
va = a.voxels(); a.close()
os.environ["TM_VOXEL_BRUTEFORCE"] = "1"
t = time.time(); b = capi.Model(ctx, model, res); b.build_voxels(); tb = time.time() - t
vb = b.voxels()
print("kd %.3fs brute %.3fs voxels %d equal %s" % (ta, tb, va.size, bool(np.array_equal(va, vb))))
