import sys, numpy as np
sys.path.insert(0, ".")
from triplet_match_amd import capi, synth
model, subset = synth.synth_model(100000, seed=1); res = synth.resolution(model)
scene = synth.synth_scene(1000000, 100000, seed=2, copies=4, model_resolution=res)
ctx = capi.Context(0)
gm = capi.Model(ctx, model, res); gm.build_voxels(); gm.build_pairs(subset)
gs = capi.Scene(ctx, scene["surfels"], scene["tangent_mask"])
pairs = synth.sample_scene_pairs(scene, gm.diameter, 300, 11, seed=13)
poses, prov, total = gs.generate_hypotheses(gm, pairs, cap=400000)
counts, scores = gs.score_hypotheses(gm, poses, 3.0)
print("pairs", len(pairs), "hyps", len(poses), "max", counts.max(), ">50k:", int((counts > 50000).sum()), ">30k:", int((counts>30000).sum()), ">10k:", int((counts>10000).sum()))
firsts = pairs[prov[:,0],0]
good = np.unique(firsts[counts>50000])
print("distinct first points with a sufficient hypothesis:", len(good), "of", len(np.unique(pairs[:,0])), "firsts; on instances:", int((scene["instance_id"][np.unique(pairs[:,0])]>=0).sum()))
