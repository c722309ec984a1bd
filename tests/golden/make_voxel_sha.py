#!/usr/bin/env python3
"""Generates tests/golden/c2_voxel_grid.json: SHA-256 of the voxel NN grid of the
BASELINE configs[1] model (synth_model(100000, seed=1)) as built by the CPU ORACLE on its own — once with its
kd-tree, once exhaustively (every voxel centre against every point, strict minimum, lowest index on ties).
tests/test_gpu_fullsize.py compares the hash of the GPU's table with it, so the full-size tie-breaking no
longer rests on the equality of two GPU kernels. The exhaustive pass is 1.7e13 distance evaluations:
about an hour on 8 cores.   python tests/golden/make_voxel_sha.py [--skip-exhaustive]"""
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


def main():
    threads = os.cpu_count() or 8
    model, _ = synth.synth_model(100_000, seed=1)
    res = synth.resolution(model)
    out = {"model": "synth_model(100000, seed=1)", "resolution": res, "threads": threads}
    for name, brute in (("kd_tree", False), ("exhaustive", True)):
        if brute and "--skip-exhaustive" in sys.argv:
            continue
        m = tmo.Model(model, res)
        t0 = time.time()
        m.build_voxels(threads=threads, brute=brute)
        v = m.voxels()
        ext, _ = m.grid()
        out["extents"] = [int(e) for e in ext]
        out["voxels"] = int(v.size)
        out["sha256_" + name] = hashlib.sha256(np.ascontiguousarray(v, np.uint32).tobytes()).hexdigest()
        out["seconds_" + name] = round(time.time() - t0, 1)
        print(name, out["sha256_" + name], out["seconds_" + name], flush=True)
        json.dump(out, open(os.path.join(ROOT, "tests", "golden", "c2_voxel_grid.json"), "w"), indent=1)
        del m, v


if __name__ == "__main__":
    main()
