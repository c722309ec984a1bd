"""Committed golden vectors (tests/golden/small_case.npz, produced by tests/golden/make_golden.py from
the CPU oracle — the reference has no fixtures of its own). CPU: the oracle still reproduces them.
GPU: the HIP path reproduces them through the C ABI without the oracle in the loop."""
import hashlib
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "small_case.npz")
DT, ACCEPT = 3.0, 0.5


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_generator_is_stable(g):
    """the fixture inputs are what synth produces today (guards the seeded generators)"""
    from triplet_match_amd import synth
    model, subset = synth.synth_model(700, seed=21)
    assert np.array_equal(bits(model), bits(g["model"])) and np.array_equal(subset, g["subset"])


def test_oracle_reproduces_golden(g):
    from oracle import tmo
    m = tmo.Model(g["model"], float(g["resolution"]))
    m.build_voxels(threads=4)
    assert m.build_pairs(g["subset"]) == 0
    s = tmo.Scene(g["scene"], g["tangent_mask"])
    ext, tv = m.grid()
    assert np.array_equal(ext, g["extents"]) and np.array_equal(bits(tv), bits(g["to_voxel"]))
    vox = m.voxels()
    assert hashlib.sha256(vox.tobytes()).digest() == g["voxel_sha256"].tobytes()
    keys, counts = m.table_keys()
    assert np.array_equal(keys, g["table_keys"]) and np.array_equal(counts, g["table_counts"])
    poses, prov = s.generate_hypotheses(m, g["pairs"])
    assert np.array_equal(prov, g["gen_prov"]) and np.array_equal(bits(poses), bits(g["gen_poses"]))
    cnt, sc, _ = s.score_hypotheses(m, g["score_poses"], ACCEPT, DT)
    assert np.array_equal(cnt, g["score_counts"]) and np.array_equal(sc, g["score_scores"])
    for h in (0, 5, len(cnt) - 1):
        e = s.project(m, g["score_poses"][h], ACCEPT, DT, subset=g["ed_subset"], early_out=True)
        assert len(e[0]) == g["ed_counts"][h] and e[3] == g["ed_saved"][h]


@pytest.mark.gpu
def test_hip_reproduces_golden(g, gpu_ctx):
    from triplet_match_amd import capi
    m = capi.Model(gpu_ctx, g["model"], float(g["resolution"]))
    m.build_voxels()
    m.build_pairs(g["subset"])
    s = capi.Scene(gpu_ctx, g["scene"], g["tangent_mask"])
    ext, tv = m.grid()
    assert np.array_equal(ext, g["extents"]) and np.array_equal(bits(tv), bits(g["to_voxel"]))
    assert m.diameter == float(g["diameter"])
    vox = m.voxels()
    assert hashlib.sha256(vox.tobytes()).digest() == g["voxel_sha256"].tobytes()
    assert np.array_equal(vox[:4096], g["voxel_head"]) and np.array_equal(vox[::997], g["voxel_stride_sample"])
    bmin, bmax = m.feature_bounds()
    assert np.array_equal(bits(bmin), bits(g["bmin"])) and np.array_equal(bits(bmax), bits(g["bmax"]))
    keys, offs, entries = m.table()
    assert np.array_equal(keys, g["table_keys"])
    assert np.array_equal(np.diff(offs.astype(np.int64)), g["table_counts"].astype(np.int64))
    assert np.array_equal(entries, g["table_entries"])
    feats, fkeys, hashes = gpu_ctx.features_batch(g["model"], g["feat_pairs"], g["bmin"], g["bmax"])
    assert np.array_equal(bits(feats), bits(g["feat_values"]))
    assert np.array_equal(fkeys, g["feat_keys"]) and np.array_equal(hashes, g["feat_hashes"])
    poses, prov, total = s.generate_hypotheses(m, g["pairs"])
    assert total == len(g["gen_prov"]) and np.array_equal(prov, g["gen_prov"])
    assert np.array_equal(bits(poses), bits(g["gen_poses"]))
    for mode in (capi.SCORE_BRUTEFORCE, capi.SCORE_AUTO):
        cnt, sc = s.score_hypotheses(m, g["score_poses"], DT, mode=mode)
        assert np.array_equal(cnt, g["score_counts"])
        assert np.allclose(sc, g["score_scores"], rtol=1e-9, atol=1e-15)
    cnt, sc, dropped, saved = s.score_hypotheses_early_drop(m, g["score_poses"], DT, ACCEPT, subset=g["ed_subset"])
    assert np.array_equal(cnt, g["ed_counts"]) and np.array_equal(saved, g["ed_saved"])
    assert np.allclose(sc, g["ed_scores"], rtol=1e-9, atol=1e-15)
    scn, mdl, score = s.extract_correspondences(m, g["truth"][0], DT)
    assert np.array_equal(scn, g["corr_scene"]) and np.array_equal(mdl, g["corr_model"])
    assert abs(score - float(g["corr_score"])) <= 1e-9 * abs(float(g["corr_score"]))
