#!/usr/bin/env python3
"""Randomised check of the k-nearest walks (tm_cloud_knn, tm_cloud_curvature, tm_cloud_resolution): eight lanes per query
(default) == one thread per query (curv_groups = 0) on trees built on the device and on the host, and == a numpy scan of
all points for the smaller clouds. Clouds: uniform, clustered, on a plane, on a line, quantised coordinates (ties), with
duplicates and non-finite points. usage (GPU box, repo root): tools/knn_sweep.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triplet_match_amd import capi  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
ctx = capi.Context(0)


def brute(points, queries, k):
    P = points.astype(np.float32)
    idx = np.full((len(queries), k), -1, np.int32)
    d2 = np.full((len(queries), k), np.inf, np.float32)
    fin = np.isfinite(P).all(axis=1)
    ids = np.nonzero(fin)[0]
    for t, q in enumerate(queries.astype(np.float32)):
        d = P[fin] - q
        v = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        order = np.lexsort((ids, v))[:k]
        idx[t, :len(order)] = ids[order]
        d2[t, :len(order)] = v[order]
    return idx, d2


bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    n = int(rng.choice([1, 2, 7, 8, 9, 33, 200, 1500, 6000, 20000]))
    kind = case % 6
    if kind == 0:
        pts = rng.random((n, 3))
    elif kind == 1:
        pts = rng.normal(size=(n, 3)) * 0.01 + rng.integers(0, 4, (n, 1))
    elif kind == 2:
        pts = np.c_[rng.random((n, 2)), np.zeros(n)]
    elif kind == 3:
        pts = np.c_[rng.random(n), np.zeros(n), np.ones(n)]
    elif kind == 4:
        pts = np.round(rng.random((n, 3)) * 6) / 6
    else:
        pts = rng.random((n, 3))
        pts[rng.integers(0, n, n // 3)] = pts[rng.integers(0, n, n // 3)]
    pts = pts.astype(np.float32)
    S = np.zeros((n, 12), np.float32)
    S[:, 0:3] = pts
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    S[:, 4:7] = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-6)
    if n > 4 and case % 5 == 0:
        S[rng.integers(0, n), rng.integers(0, 3)] = np.nan
    k = int(rng.choice([1, 2, 5, 16, 30, 31, 32]))
    queries = np.concatenate([pts[rng.integers(0, n, 40)], rng.random((10, 3)).astype(np.float32)])
    queries = queries[np.isfinite(queries).all(axis=1)]
    got = {}
    for host in (0, 1):
        for groups in (1, 0):
            ctx.set_option("model_host_build", host)
            ctx.set_option("curv_groups", groups)
            gc = capi.Cloud(ctx, S)
            idx, d2, cnt = gc.knn(queries, k)
            curv = ctx.cloud_curvature(S, k)
            res = ctx.cloud_resolution(S)[1] if np.isfinite(S[:, 0:3]).all(axis=1).sum() >= 2 else np.zeros(1, np.float32)
            got[(host, groups)] = (idx, d2.view(np.uint32), cnt, curv.view(np.uint32), res.view(np.uint32))
            gc.close()
    ref = got[(0, 1)]
    ok = all(all(np.array_equal(a, b) for a, b in zip(ref, g)) for g in got.values())
    if ok and n <= 6000:
        bi, bd = brute(S[:, 0:3], queries, k)
        ok = np.array_equal(ref[0], bi) and np.array_equal(ref[1], bd.view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH case %d: n %d kind %d k %d" % (case, n, kind, k), flush=True)
ctx.set_option("model_host_build", 0)
ctx.set_option("curv_groups", 1)
print("%d cases, %d mismatches" % (n_cases, bad))
