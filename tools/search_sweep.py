#!/usr/bin/env python3
"""End-to-end search sweep: find_all over the HIP backend must equal find_all over the oracle backend
(candidates, correspondences, transforms) for varying clouds, with the early-drop rule on and off.
usage: tools/search_sweep.py [n_cases]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_cpp_dropin import _compile, _oracle_link  # noqa: E402
from triplet_match_amd import synth  # noqa: E402


def fixture(path, seed, model_n, scene_n, copies):
    model, _ = synth.synth_model(model_n, seed=seed)
    res = synth.resolution(model)
    scene = synth.synth_scene(scene_n, model_n, seed=seed + 1, copies=copies, model_resolution=res)
    with open(path, "wb") as f:
        f.write(np.uint32(model.shape[0]).tobytes())
        f.write(model.tobytes())
        f.write(np.uint32(scene["surfels"].shape[0]).tobytes())
        f.write(scene["surfels"].tobytes())
        f.write(np.float32(res).tobytes())
        f.write(np.uint32(copies).tobytes())
        for T in scene["poses_scene_to_model"]:
            f.write(synth.pose_colmajor(T).tobytes())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    exe = _compile("test_search", _oracle_link())
    rng = np.random.default_rng(77)
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for k in range(n):
            model_n = int(rng.integers(800, 2500))
            copies = int(rng.integers(1, 4))
            scene_n = int(model_n * copies * 1.3 + rng.integers(2000, 12000))
            fx = os.path.join(d, "c%d.bin" % k)
            fixture(fx, 100 + 3 * k, model_n, scene_n, copies)
            for mode in ("both", "both_noearlydrop"):
                r = subprocess.run([exe, fx, mode], capture_output=True, text=True, timeout=1200)
                ok = "ok   hip backend == oracle backend" in r.stdout  # parity; detection quality is reported apart
                found = r.stdout.count("corrs ") // 2
                detect = "all planted poses" if r.returncode == 0 else "search misses / false positives (both backends alike)"
                print("case %2d (%d model, %d scene, %d copies) %-17s %s (%d candidates; %s)"
                      % (k, model_n, scene_n, copies, mode, "ok" if ok else "MISMATCH", found, detect), flush=True)
                if not ok:
                    bad += 1
                    print(r.stdout[-1500:], r.stderr[-500:])
    print("%d cases, %d mismatches" % (n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
