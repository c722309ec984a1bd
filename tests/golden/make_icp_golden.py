#!/usr/bin/env python3
"""Generates tests/golden/icp_sets.npz: correspondence sets and the rigid transform that best maps the scene
points onto the model points, computed here with numpy's SVD (Umeyama / Kabsch without scale, determinant
correction) in float64 — independent of the product's Horn/Jacobi solver (search::detail::rigid_fit) and of
the oracle. The reference uses Eigen::umeyama(scn, mdl, false) (include/impl/scene.hpp:385-393).
Run from the repo root:  python tests/golden/make_icp_golden.py"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def umeyama_noscale(src, dst):
    """T (4x4, float64) minimising sum |dst - (R src + t)|^2 over rotations R (det +1)."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    cs, cd = src.mean(0), dst.mean(0)
    H = (dst - cd).T @ (src - cs)  # sum (d - cd)(s - cs)^T
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = cd - R @ cs
    return T


def rot(axis, angle):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def main():
    rng = np.random.default_rng(20260402)
    cases = []

    def add(name, src, R, t, noise):
        src = np.asarray(src, np.float64)
        dst = src @ R.T + t + rng.normal(scale=noise, size=src.shape)
        # the product works on float32 points: the golden answer is for exactly those values
        s32, d32 = src.astype(np.float32), dst.astype(np.float32)
        cases.append((name, s32, d32, umeyama_noscale(s32, d32)))

    for k in range(8):  # generic clouds, increasing noise, 8..64 points
        n = int(rng.integers(8, 65))
        add("generic_%d" % k, rng.normal(size=(n, 3)), rot(rng.normal(size=3), rng.uniform(0, np.pi)),
            rng.uniform(-3, 3, size=3), 1e-3 * k)
    add("three_points", rng.normal(size=(3, 3)), rot([1, 2, 3], 0.7), [0.5, -1.0, 2.0], 0.0)
    add("four_points_noisy", rng.normal(size=(4, 3)), rot([0, 1, 0], 2.0), [1.0, 1.0, 1.0], 1e-2)
    add("identity", rng.normal(size=(32, 3)), np.eye(3), [0, 0, 0], 0.0)
    add("tiny_rotation", rng.normal(size=(40, 3)), rot([1, 0, 0], 1e-4), [1e-3, 0, 0], 1e-5)
    add("half_turn", rng.normal(size=(40, 3)), rot([0, 0, 1], np.pi), [0.2, 0.1, -0.3], 1e-4)
    add("almost_half_turn", rng.normal(size=(40, 3)), rot([1, 1, 0], np.pi - 1e-3), [2.0, -2.0, 0.5], 1e-4)
    flat = rng.normal(size=(48, 3)) * np.array([1.0, 1.0, 0.05])  # nearly coplanar
    add("nearly_planar", flat, rot([1, 0, 1], 1.1), [0.3, 0.3, 0.3], 1e-3)
    add("exactly_planar", flat * np.array([1.0, 1.0, 0.0]), rot([0, 1, 1], 0.4), [-1.0, 0.0, 1.0], 0.0)
    line = rng.normal(size=(48, 3)) * np.array([1.0, 0.05, 0.05])  # elongated: weak about its axis
    add("elongated", line, rot([0.2, 1, 0], 0.9), [1.5, -0.5, 0.0], 1e-4)
    add("far_from_origin", rng.normal(size=(30, 3)) + np.array([4.0, -3.0, 2.0]), rot([3, 1, 2], 1.9),
        [-4.0, 4.0, 1.0], 1e-3)
    add("small_cloud", 1e-2 * rng.normal(size=(30, 3)), rot([1, 1, 1], 0.5), [0.1, 0.2, 0.3], 1e-5)
    add("reflection_trap", rng.normal(size=(5, 3)) * np.array([1, 1, 1e-3]), rot([0, 0, 1], 0.3), [0, 0, 0], 2e-3)

    nmax = max(c[1].shape[0] for c in cases)
    src = np.zeros((len(cases), nmax, 3), np.float32)
    dst = np.zeros_like(src)
    cnt = np.zeros(len(cases), np.uint32)
    T = np.zeros((len(cases), 4, 4), np.float64)
    for i, (_, s, d, t) in enumerate(cases):
        cnt[i] = s.shape[0]
        src[i, :cnt[i]], dst[i, :cnt[i]], T[i] = s, d, t
    np.savez_compressed(os.path.join(HERE, "icp_sets.npz"), names=np.array([c[0] for c in cases]), count=cnt,
                        scene=src, model=dst, transform=T)
    print("wrote %d sets" % len(cases))


if __name__ == "__main__":
    main()
