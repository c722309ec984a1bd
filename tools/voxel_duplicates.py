#!/usr/bin/env python3
"""How much of the voxel NN table is redundant near the surface? For the 2x2x2-voxel blocks whose eight voxel centres
lie within `reach` (default: dist_thres 3 x resolution + one cell diagonal) of the model — the only part of the table the
scoring kernels ever gather from — count the blocks whose eight entries are all equal, and the mean number of distinct
entries per block; the same for 4x4x4 bricks. CPU only (the oracle's table of a 20k-point model; the cell is half the
point spacing at every model size, so the rates do not depend on it).  usage: python tools/voxel_duplicates.py [n_points]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tmo  # noqa: E402
from triplet_match_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
model, _ = synth.synth_model(n, seed=1)
res = synth.resolution(model)
m = tmo.Model(model, res)
m.build_voxels(threads=8)
ext, to_voxel = m.grid()
vox = m.voxels().reshape(ext[2], ext[1], ext[0])  # [k][j][i]
T = np.asarray(to_voxel, np.float64).reshape(4, 4).T
scale = np.array([T[0, 0], T[1, 1], T[2, 2]])
trans = T[:3, 3]
kk, jj, ii = np.meshgrid(np.arange(ext[2]), np.arange(ext[1]), np.arange(ext[0]), indexing="ij")
centre = (np.stack([ii, jj, kk], -1) - trans) / scale
d = np.linalg.norm(centre - model[vox, 0:3].astype(np.float64), axis=-1)
cell = float(np.max(1.0 / scale))
reach = 3.0 * res + np.sqrt(3.0) * cell
print("model %d points, resolution %.5f, grid %s = %.2e voxels, %.1f %% within reach of the surface"
      % (n, res, list(ext), vox.size, 100.0 * float((d <= reach).mean())))
for B in (2, 4):
    e = [x - x % B for x in (ext[2], ext[1], ext[0])]
    v = vox[:e[0], :e[1], :e[2]].reshape(e[0] // B, B, e[1] // B, B, e[2] // B, B).transpose(0, 2, 4, 1, 3, 5).reshape(-1, B ** 3)
    dn = d[:e[0], :e[1], :e[2]].reshape(e[0] // B, B, e[1] // B, B, e[2] // B, B).transpose(0, 2, 4, 1, 3, 5).reshape(-1, B ** 3)
    near = dn.min(axis=1) <= reach
    vs = np.sort(v[near], axis=1)
    distinct = 1 + (np.diff(vs, axis=1) != 0).sum(axis=1)
    print("%dx%dx%d blocks within reach: %d; all entries equal: %.1f %%; distinct entries per block: mean %.2f of %d"
          % (B, B, B, int(near.sum()), 100.0 * float((distinct == 1).mean()), float(distinct.mean()), B ** 3))
