#!/usr/bin/env python3
"""Census of the one place where oracle and kernels knowingly differ from the reference's BINARY: the reference calls
glibc's atan2f (include/impl/feature.hpp:7), the project its own Cephes-style atan2f (<= 1.1 ulp apart, so that host,
device and oracle agree bit for bit). How many descriptor keys does that change?

  (1) every ordered pair of the configs[1] model's tangent subset that passes the model's pair filters
      (include/impl/model.hpp:104-112) — the pairs behind the 1.4e7-entry key table;
  (2) 1,000,000 scene sample pairs of the configs[1] scene (tangent points within the model's diameter window).

For each pair both angles are formed with both functions and binned with angle_step = 10 degrees
(src/discretize.cpp:26-30). Writes tests/golden/atan2_census.json: counts, the largest difference, and the differing
pairs (tests/test_oracle_kats.py checks the committed counts against a fresh census on a smaller model and the
file's own consistency). CPU only, ~20 s.   usage: python tests/golden/make_atan2_census.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tmo  # noqa: E402
from triplet_match_amd import synth  # noqa: E402


def census(model_points, scene_points, n_scene_pairs, seed=13):
    angle_step = float(np.float32(10.0) * np.float32(np.pi) / np.float32(180.0))
    model, subset = synth.synth_model(model_points, seed=1)
    res = synth.resolution(model)
    scene = synth.synth_scene(scene_points, model_points, seed=2, copies=4, model_resolution=res)
    finite = model[np.isfinite(model[:, 0:3]).all(axis=1)]
    diameter = float(np.linalg.norm(finite[:, 0:3].max(axis=0) - finite[:, 0:3].min(axis=0)).astype(np.float32))
    out = {"angle_step": angle_step, "model_points": model_points, "scene_points": scene_points,
           "model_tangent_subset": int(len(subset))}
    m = tmo.atan2_census(model, angle_step, subset=subset, lower=0.2 * diameter, upper=1.0 * diameter)
    out["model_pairs"] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in m.items()}
    per_first = 24
    pairs = synth.sample_scene_pairs(scene, diameter, max(1, n_scene_pairs // per_first), per_first, seed=seed)
    s = tmo.atan2_census(scene["surfels"], angle_step, ij=pairs)
    out["scene_pairs"] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in s.items()}
    return out


if __name__ == "__main__":
    res = {"full_size": census(100_000, 1_000_000, 1_000_000), "small": census(5_000, 50_000, 50_000)}
    path = os.path.join(ROOT, "tests", "golden", "atan2_census.json")
    json.dump(res, open(path, "w"), indent=1)
    for k, v in res.items():
        print(k, {a: {b: c for b, c in d.items() if b != "differing_ij"} for a, d in v.items() if isinstance(d, dict)})
