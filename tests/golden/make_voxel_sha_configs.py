#!/usr/bin/env python3
"""Generates tests/golden/c345_voxel_grids.json: SHA-256 of the voxel NN grids of the
models of BASELINE configs[2], [3] and [4] as the CPU ORACLE builds them on its own, with no GPU result anywhere in
the chain:

  C3  synth_pipe_section(50000, seed=3) projected by the oracle's cylinder_traits functions (init_from_samples of
      the sample pair tests/test_gpu_fullsize_traits.py draws, project / tangent / normal per point), then
      tmo.Model.build_voxels in uvw space;
  C4  synth_wall_panel(15000, seed=5) through the oracle's plane2_traits functions likewise;
  C5  models 0, 15 and 31 of synth_multi_model_scene(4000000, 32, seed=7).

tests/test_gpu_fullsize_traits.py / test_gpu_fullsize_multimodel.py assert that the GPU's projected model and voxel table
hash to these values BEFORE they hand the table to the oracle's project_ (om.set_voxels), so those tests no longer
rest on a table no independent implementation has built.   python tests/golden/make_voxel_sha_configs.py  (minutes)"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import tmo  # noqa: E402
from triplet_match_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "c345_voxel_grids.json")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pick_cylinder_samples(s, sel, rng):
    """the draw of tests/test_gpu_fullsize_traits.py:_pick_cylinder_samples"""
    idx = np.nonzero(sel)[0]
    a = int(idx[rng.integers(len(idx))])
    for _ in range(1000):
        b = int(idx[rng.integers(len(idx))])
        if abs(float(s[a, 4:7] @ s[b, 4:7])) < 0.3:
            return [a, b]
    raise AssertionError("no sample pair")


def pick_plane_samples(s, sel, rng):
    idx = np.nonzero(sel)[0]
    for _ in range(1000):
        a, b, c = (int(idx[k]) for k in rng.integers(len(idx), size=3))
        d1, d2 = s[b, 0:3] - s[a, 0:3], s[c, 0:3] - s[a, 0:3]
        if np.linalg.norm(np.cross(d1, d2)) > 0.5 * np.linalg.norm(d1) * np.linalg.norm(d2) > 0.05:
            return [a, b, c]
    raise AssertionError("no sample triple")


def project_cloud_oracle(kind, state, surfels):
    """every point through the oracle's scalar traits functions (tests/test_gpu_parity.py:_project_cloud_oracle)"""
    out = np.zeros_like(surfels)
    valid = np.zeros(len(surfels), np.int32)
    for i, s in enumerate(surfels):
        uvw = tmo.traits_project(kind, state, s[0:3])
        if uvw is None:
            continue
        valid[i] = 1
        out[i, 0:3] = uvw
        out[i, 3] = 1
        if np.linalg.norm(s[9:12]) > 0:
            out[i, 9:12] = tmo.traits_tangent(kind, state, s)
            out[i, 4:7] = tmo.traits_normal(kind, state, s)
        else:
            g = np.asarray(state)[5:21].reshape(4, 4).T[:3, :3]
            out[i, 4:7] = g.astype(np.float32) @ s[4:7]
    return out, valid


def grid_entry(cloud, res, threads, extra=None):
    t0 = time.time()
    m = tmo.Model(cloud, res)
    m.build_voxels(threads=threads)
    v = m.voxels()
    ext, tv = m.grid()
    e = dict(points=int(len(cloud)), cloud_sha256=sha(np.ascontiguousarray(cloud, np.float32)),
             positions_sha256=sha(np.ascontiguousarray(np.asarray(cloud)[:, 0:3], np.float32)),
             resolution=float(res), extents=[int(x) for x in ext], voxels=int(v.size),
             voxel_sha256=sha(np.ascontiguousarray(v, np.uint32)), seconds=round(time.time() - t0, 1))
    if extra:
        e.update(extra)
    return e


def projected_model(kind, model, msamples, threshold):
    ms = np.zeros(tmo.STATE_FLOATS, np.float32)
    ms[1] = threshold
    mstate = tmo.traits_init_from_samples(kind, ms, np.concatenate([model[i] for i in msamples]))
    assert mstate is not None
    mu, mv = project_cloud_oracle(kind, mstate, model)
    return mu[mv == 1], mstate


def main():
    threads = os.cpu_count() or 8
    out = {}

    def save(name, e):
        out[name] = e
        print(name, e["voxel_sha256"], e["voxels"], e["seconds"], flush=True)
        json.dump(out, open(OUT, "w"), indent=1)

    model, _ = synth.synth_pipe_section(50_000, seed=3)
    ms = pick_cylinder_samples(model, np.ones(len(model), bool), np.random.default_rng(2))
    mu, st = projected_model(tmo.CYLINDER, model, ms, 0.05)
    save("c3_pipe_section_cylinder", grid_entry(mu, synth.resolution(mu), threads,
                                                dict(model="synth_pipe_section(50000, seed=3)", msamples=ms, radius=float(st[0]))))
    model, _ = synth.synth_wall_panel(15_000, seed=5)
    ms = pick_plane_samples(model, np.ones(len(model), bool), np.random.default_rng(3))
    mu, st = projected_model(tmo.PLANE2, model, ms, 0.05)
    save("c4_wall_panel_plane2", grid_entry(mu, synth.resolution(mu), threads,
                                            dict(model="synth_wall_panel(15000, seed=5)", msamples=ms)))
    models, _ = synth.synth_multi_model_scene(4_000_000, 32, seed=7)
    for k in (0, 15, 31):
        M, sub, res = models[k]
        save("c5_model_%d" % k, grid_entry(M, res, threads, dict(model="synth_multi_model_scene(4000000, 32, seed=7)[%d]" % k)))


if __name__ == "__main__":
    main()
