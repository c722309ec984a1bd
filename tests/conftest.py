"""Shared fixtures. GPU tests are marked `gpu` and call the product only through the C ABI
(triplet_match_amd.capi -> libtm_hip.so); the oracle (oracle/) is the checker, never the thing
under test in a gpu test."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


DIST_THRES = 3.0
ACCEPT_PROB = 0.5


@pytest.fixture(scope="session")
def small_case():
    """2k-pt model, 20k-pt scene with two planted instances (seeded)."""
    from triplet_match_amd import synth
    model, subset = synth.synth_model(2000, seed=1)
    res = synth.resolution(model)
    scene = synth.synth_scene(20000, 2000, seed=2, copies=2, model_resolution=res)
    return dict(model=model, subset=subset, resolution=res, scene=scene)


@pytest.fixture(scope="session")
def oracle_small(small_case):
    from oracle import tmo
    m = tmo.Model(small_case["model"], small_case["resolution"])
    m.build_voxels(threads=8)
    rc = m.build_pairs(small_case["subset"])
    assert rc == 0
    s = tmo.Scene(small_case["scene"]["surfels"], small_case["scene"]["tangent_mask"])
    return dict(model=m, scene=s)


@pytest.fixture(scope="session")
def gpu_ctx():
    """HIP context on device 0. Fails (does not skip) when the library or the device is missing:
    a gpu-marked test that cannot reach the HIP path must not pass."""
    from triplet_match_amd import capi
    return capi.Context(0)


@pytest.fixture(scope="session")
def gpu_small(gpu_ctx, small_case):
    from triplet_match_amd import capi
    m = capi.Model(gpu_ctx, small_case["model"], small_case["resolution"])
    m.build_voxels()
    m.build_pairs(small_case["subset"])
    s = capi.Scene(gpu_ctx, small_case["scene"]["surfels"], small_case["scene"]["tangent_mask"])
    return dict(model=m, scene=s)


def sample_pairs(small_case, oracle_model, n_first=24, per_first=6, seed=13):
    from triplet_match_amd import synth
    return synth.sample_scene_pairs(small_case["scene"], oracle_model.diameter, n_first, per_first,
                                    seed=seed)


def random_poses_near_truth(small_case, n, seed=5, jitter=0.02):
    """Ground-truth scene->model poses with small perturbations plus a few random ones."""
    from triplet_match_amd import synth
    rng = np.random.default_rng(seed)
    truth = small_case["scene"]["poses_scene_to_model"]
    out = []
    for k in range(n):
        T = truth[k % len(truth)].copy()
        if k >= len(truth):
            if k % 5 == 4:
                T = np.eye(4)
                T[:3, :3] = synth.random_rotation(rng)
                T[:3, 3] = rng.normal(size=3)
            else:
                dR = synth.random_rotation(rng)
                a = jitter * rng.random()
                R = np.eye(3) * (1 - a) + dR * a
                u, _, vt = np.linalg.svd(R)
                R = u @ vt
                T[:3, :3] = R @ T[:3, :3]
                T[:3, 3] = R @ T[:3, 3] + rng.normal(size=3) * jitter * 0.2
        out.append(synth.pose_colmajor(T))
    return np.array(out, dtype=np.float32)
