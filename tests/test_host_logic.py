"""Independent checks of the host-side steps after scoring:
 * search::detail::rigid_fit (Horn / Jacobi, what the ICP loop solves with) against numpy SVD-Umeyama golden
   vectors (tests/golden/icp_sets.npz, made by tests/golden/make_icp_golden.py) — the reference calls
   Eigen::umeyama(scn, mdl, false), include/impl/scene.hpp:385-393;
 * search::arbitrate against a hand-worked example (include/impl/scene.hpp:193-237);
 * GPU: tm_icp_accumulate's 15 fixed-point sums against the same sums formed with numpy integers over the
   correspondences, and the pose they lead to against numpy's SVD solution."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "_build")
GOLDEN = os.path.join(ROOT, "tests", "golden", "icp_sets.npz")


def _exe():
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "test_host_logic")
    src = os.path.join(ROOT, "tests", "cpp", "test_host_logic.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wno-comment", "-I" + os.path.join(ROOT, "include"),
                           src, "-o", out])
    return out


def _rigid_fit(tmp_path, sets):
    """sets: list of (scene[n,3] f32, model[n,3] f32) -> list of 4x4 float32 (row/col as math, not storage)"""
    fin, fout = str(tmp_path / "sets.bin"), str(tmp_path / "poses.bin")
    with open(fin, "wb") as f:
        f.write(np.uint32(len(sets)).tobytes())
        for s, m in sets:
            f.write(np.uint32(len(s)).tobytes())
            f.write(np.ascontiguousarray(s, np.float32).tobytes())
            f.write(np.ascontiguousarray(m, np.float32).tobytes())
    subprocess.check_call([_exe(), "rigid_fit", fin, fout])
    raw = np.fromfile(fout, np.float32).reshape(len(sets), 4, 4)
    return [r.T.astype(np.float64) for r in raw]  # column-major storage -> math


def test_rigid_fit_matches_numpy_umeyama_golden(tmp_path):
    g = np.load(GOLDEN)
    sets = [(g["scene"][i, :g["count"][i]], g["model"][i, :g["count"][i]]) for i in range(len(g["count"]))]
    assert len(sets) >= 20
    got = _rigid_fit(tmp_path, sets)
    for name, T, want in zip(g["names"], got, g["transform"]):
        err = np.abs(T - want).max()
        assert err <= 1e-6, "%s: max |rigid_fit - umeyama| = %.3g" % (name, err)
        assert abs(np.linalg.det(T[:3, :3]) - 1.0) < 1e-5, name  # a rotation, never a reflection


def test_golden_script_reproduces_the_committed_vectors(tmp_path):
    """the generating script is committed next to the data: running it again gives the same file content"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_icp_golden", os.path.join(ROOT, "tests", "golden",
                                                                                  "make_icp_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(GOLDEN)
    for i in (0, 5, 13, 19):
        n = g["count"][i]
        T = mod.umeyama_noscale(g["scene"][i, :n], g["model"][i, :n])
        assert np.allclose(T, g["transform"][i], atol=1e-12)


def test_arbitration_hand_worked_example():
    r = subprocess.run([_exe(), "arbitrate"], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0 and "PASSED" in r.stdout and "FAIL" not in r.stdout


@pytest.mark.gpu
def test_icp_accumulate_sums_and_pose(gpu_ctx, gpu_small, small_case, tmp_path):
    """tm_icp_accumulate (finish_find + the ICP sums of icp_correlation, opencl/icp.cl:55-86, in 2^-32 fixed
    point) == the same integers formed on the host from the correspondences tm_extract_correspondences returns;
    the pose those sums give (rigid_fit) == numpy's SVD solution on the correspondences (1e-6)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_icp_golden", os.path.join(ROOT, "tests", "golden",
                                                                                  "make_icp_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from triplet_match_amd import synth
    gm, gs = gpu_small["model"], gpu_small["scene"]
    model, scene = small_case["model"], small_case["scene"]["surfels"]
    rng = np.random.default_rng(11)
    for k, T in enumerate(small_case["scene"]["poses_scene_to_model"]):
        # a slightly wrong pose: what an ICP pass starts from
        dR = mod.rot(rng.normal(size=3), 0.01)
        P = np.eye(4)
        P[:3, :3] = dR @ T[:3, :3]
        P[:3, 3] = dR @ T[:3, 3] + rng.normal(size=3) * 0.5 * small_case["resolution"]
        pose = synth.pose_colmajor(P).astype(np.float32)
        n, sums = gs.icp_accumulate(gm, pose, 6.0)
        sc, mc, _ = gs.extract_correspondences(gm, pose, 6.0)
        assert n == len(sc) and n > 200
        s = scene[sc, 0:3].astype(np.float64)
        m = model[mc, 0:3].astype(np.float64)
        S = 4294967296.0
        want = np.zeros(15, np.int64)
        want[0:3] = np.trunc(s * S).astype(np.int64).sum(0)
        want[3:6] = np.trunc(m * S).astype(np.int64).sum(0)
        for a in range(3):
            for b in range(3):
                want[6 + 3 * a + b] = np.trunc((s[:, a] * m[:, b]) * S).astype(np.int64).sum()
        assert np.array_equal(np.asarray(sums, np.int64), want), "pose %d: fixed-point sums differ" % k
        got = _rigid_fit(tmp_path, [(scene[sc, 0:3], model[mc, 0:3])])[0]
        ref = mod.umeyama_noscale(scene[sc, 0:3], model[mc, 0:3])
        assert np.abs(got - ref).max() <= 1e-6


def _sums(s, m):
    """the 15 fixed-point sums of tm_icp_accumulate formed with numpy integers"""
    s, m = np.asarray(s, np.float32).astype(np.float64), np.asarray(m, np.float32).astype(np.float64)
    S = 4294967296.0
    v = np.zeros(15, np.int64)
    v[0:3] = np.trunc(s * S).astype(np.int64).sum(0)
    v[3:6] = np.trunc(m * S).astype(np.int64).sum(0)
    for a in range(3):
        for b in range(3):
            v[6 + 3 * a + b] = np.trunc((s[:, a] * m[:, b]) * S).astype(np.int64).sum()
    return v


def test_library_rigid_fit_equals_search_hpp_and_golden(tmp_path):
    """tm_rigid_fit (csrc/tm_icp_math.h: what the device-side ICP loop runs, here on the host) gives bit for bit
    the pose of search::detail::rigid_fit on the same correspondences, and both match the numpy SVD golden."""
    from triplet_match_amd import capi
    g = np.load(GOLDEN)
    sets = [(g["scene"][i, :g["count"][i]], g["model"][i, :g["count"][i]]) for i in range(len(g["count"]))]
    hpp = _rigid_fit(tmp_path, sets)
    for (s, m), T_hpp, want, name in zip(sets, hpp, g["transform"], g["names"]):
        T = capi.rigid_fit(_sums(s, m), len(s)).reshape(4, 4).T.astype(np.float64)
        assert np.array_equal(T, T_hpp), name
        assert np.abs(T - want).max() <= 1e-6, name
    with pytest.raises(capi.TmError):
        capi.rigid_fit(np.zeros(15, np.int64), 2)


@pytest.mark.gpu
def test_icp_refine_on_the_device_equals_the_host_loop(gpu_ctx, gpu_small, small_case):
    """tm_icp_refine (passes, fits and the stop test on the device) == the loop of icp_ (reference
    include/impl/scene.hpp:376-400) driven from the host over tm_icp_accumulate + tm_rigid_fit: same pose bits,
    same number of correspondences and iterations, for several starting poses and iteration limits."""
    from triplet_match_amd import capi, synth
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_icp_golden", os.path.join(ROOT, "tests", "golden",
                                                                                  "make_icp_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gm, gs = gpu_small["model"], gpu_small["scene"]
    rng = np.random.default_rng(19)
    dt2 = 6.0
    moved = 0
    batch = {0: [], 1: [], 3: [], 10: []}  # (start, pose, count, iterations) per iteration limit
    for k, T in enumerate(small_case["scene"]["poses_scene_to_model"]):
        for trial, (angle, shift) in enumerate([(0.01, 0.5), (0.03, 1.5), (0.0, 0.0), (1.0, 50.0)]):
            dR = mod.rot(rng.normal(size=3), angle)
            P = np.eye(4)
            P[:3, :3] = dR @ T[:3, :3]
            P[:3, 3] = dR @ T[:3, 3] + rng.normal(size=3) * shift * small_case["resolution"]
            start = synth.pose_colmajor(P).astype(np.float32)
            for max_it in (0, 1, 3, 10):
                # host loop
                best_t = start.copy()
                best_n, best = gs.icp_accumulate(gm, best_t, dt2)
                it = 0
                while max_it > 0:
                    if best_n < 3:
                        break
                    trans = capi.rigid_fit(best, best_n)
                    n, sums = gs.icp_accumulate(gm, trans, dt2)
                    if n < best_n:
                        break
                    best_n, best, best_t = n, sums, trans
                    it += 1
                    if it == max_it:
                        break
                pose, cnt, iters = gs.icp_refine(gm, start, dt2, max_it)
                assert cnt == best_n and iters == it, (k, trial, max_it)
                assert np.array_equal(pose.view(np.uint32), np.asarray(best_t, np.float32).view(np.uint32)), (k, trial, max_it)
                moved += int(it > 0)
                batch[max_it].append((start, pose, cnt, iters))
    assert moved >= 6
    # tm_icp_refine_batch: the same candidates side by side (loops of different lengths in one batch) = each alone
    for max_it, rows in batch.items():
        poses, cnts, its = gs.icp_refine_batch(gm, np.stack([r[0] for r in rows]), dt2, max_it)
        for k, (_, pose, cnt, iters) in enumerate(rows):
            assert cnts[k] == cnt and its[k] == iters, (max_it, k)
            assert np.array_equal(poses[k].view(np.uint32), np.asarray(pose, np.float32).view(np.uint32)), (max_it, k)
