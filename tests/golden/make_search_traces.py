#!/usr/bin/env python3
"""Generates tests/golden/search_trace_*.json.gz: chunk-by-chunk traces of find_all_parallel / find_parallel /
find_in_subset as oracle/ref_search.py (the reference text restated in Python/numpy, no code shared with
include/triplet_match/search.hpp or oracle/tm_oracle.cpp) runs them.

The reference holds no tests or vectors and cannot be built here, so these traces are what
pins the product's sampling loops (rows a12 / a13): tests/test_search_traces.py runs search.hpp over the CPU oracle
backend (CPU test) and over the HIP backend (-m gpu) and requires every entry below to be reproduced: RANSAC bounds,
the sampled first points in order, chunk split, per first point the neighbourhood size, inner bound, draws and valid
second points, the first sufficient hypothesis of every chunk (pose bit for bit), `cons`, the candidates after ICP and
the accepted matches with their correspondence lists.

Run from the repo root (minutes on 8 cores):  python tests/golden/make_search_traces.py [case ...]
"""
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_search as rs  # noqa: E402
from oracle import tmo  # noqa: E402
from triplet_match_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (fixture arguments of tests/test_cpp_dropin.py:_fixture, search parameter sets)
CASES = {
    "2k_20k": dict(model_n=2000, scene_n=20000, copies=2, instance_points=None),
    "5k_5k": dict(model_n=5000, scene_n=5000, copies=1, instance_points=3800),  # BASELINE configs[0]
}
PARAM_SETS = [
    dict(name="icp0", max_icp_iterations=0, early_drop=True, workers=4),
    dict(name="icp2", max_icp_iterations=2, early_drop=True, workers=4),
    dict(name="icp0_noearlydrop_w7", max_icp_iterations=0, early_drop=False, workers=7),
]
DIST_THRES, ACCEPT, EARLY_OUT = 3.0, 0.5, 0.9
SAMPLE = dict(min_diameter_factor=0.2, max_diameter_factor=1.0, force_up=False)


def build_inputs(case):
    model, _ = synth.synth_model(case["model_n"], seed=1)
    res = synth.resolution(model)
    scene = synth.synth_scene(case["scene_n"], case["model_n"], seed=2, copies=case["copies"], model_resolution=res,
                              instance_points=case["instance_points"])
    return model, np.float32(res), scene["surfels"]


def run_case(name):
    case = CASES[name]
    model, res, scene = build_inputs(case)
    t0 = time.time()
    # row f1 (curvature(30, i)) is pinned on its own (tests/golden/curvature_20k.npz); taken from the C++ oracle here
    mcurv = tmo.cloud_curvature(model, 30)
    scurv = tmo.cloud_curvature(scene, 30)
    m = rs.Model(model, res, mcurv)
    print("[%s] model restated in %.1f s: %d voxels, subset %d, %d table entries" %
          (name, time.time() - t0, len(m.voxels), len(m.subset), m.pair_count), flush=True)
    s = rs.Scene(scene, curvature=scurv)
    out = dict(
        case=dict(name=name, **case, model_sha256=rs.sha256_of(model), scene_sha256=rs.sha256_of(scene),
                  resolution_bits=rs.bits([res])[0], dist_thres=DIST_THRES, accept_prob=ACCEPT, early_out_factor=EARLY_OUT,
                  **SAMPLE, seed=13),
        model=dict(diameter_bits=rs.bits([m.diameter])[0], extents=[int(e) for e in m.extents], to_voxel=rs.bits(m.to_voxel),
                   voxel_sha256=rs.sha256_of(m.voxels), subset_fnv=rs.fnv1a64(m.subset), n_subset=int(len(m.subset)),
                   bmin=rs.bits(m.bmin), bmax=rs.bits(m.bmax), pair_count=m.pair_count, table_fnv=m.table_fingerprint()),
        scene=dict(n_tangent=int(s.tangent_mask.sum()), tangent_mask_fnv=rs.fnv1a64(s.tangent_mask.astype(np.int32))),
        searches=[])
    for ps in PARAM_SETS:
        t0 = time.time()
        results, trace = s.find_all_parallel(m, DIST_THRES, ACCEPT, EARLY_OUT, SAMPLE, ps["max_icp_iterations"],
                                             workers=ps["workers"], seed=13, early_drop=ps["early_drop"])
        trace["params"] = ps
        trace["n_results"] = len(results)
        out["searches"].append(trace)
        print("[%s/%s] %d rounds, %d matches, %.1f s" % (name, ps["name"], len(trace["rounds"]), len(results),
                                                          time.time() - t0), flush=True)
    if name == "2k_20k":
        # the reference keeps calling project_ after a chunk has its candidate; those calls change nothing (see
        # ref_search.Scene.find_all_parallel): shown once by running them all and comparing what the search returns
        t0 = time.time()
        ps = PARAM_SETS[0]
        results, trace = s.find_all_parallel(m, DIST_THRES, ACCEPT, EARLY_OUT, SAMPLE, ps["max_icp_iterations"],
                                             workers=ps["workers"], seed=13, early_drop=ps["early_drop"], dead_work=True)
        ref = out["searches"][0]
        same = len(trace["rounds"]) == len(ref["rounds"])
        for a, b in zip(trace["rounds"], ref["rounds"]):
            same = same and a["accepted"] == b["accepted"] and a["candidates"] == b["candidates"] and a["outer"] == b["outer"]
            for ca, cb in zip(a["chunks"], b["chunks"]):
                same = same and ca["hit"] == cb["hit"] and ca["cons"] == cb["cons"]
                for fa, fb in zip(ca["firsts"], cb["firsts"]):
                    same = same and all(fa[k] == fb[k] for k in ("i", "nn", "inner_bound", "draws", "js", "dead"))
        out["dead_work_changes_nothing"] = bool(same)
        print("[%s] dead work run: same = %s, %.1f s" % (name, same, time.time() - t0), flush=True)
        assert same
    path = os.path.join(HERE, "search_trace_%s.json.gz" % name)
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(out, separators=(",", ":")).encode())
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    for nm in (sys.argv[1:] or list(CASES)):
        run_case(nm)
