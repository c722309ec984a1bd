"""N > 1 path on CPU: two gloo ranks shard the sample pairs (triplet_match_amd.dist), score their own
hypotheses (with the CPU oracle standing in for the GPU kernels — no GPU here), exchange the 96-byte
best records with one all-gather and select the global best with tm_best_select. Both ranks must agree
with each other and with the single-process argmax over all hypotheses."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from oracle import tmo
    from triplet_match_amd import build as tm_build
    from triplet_match_amd import dist as tmd
    from triplet_match_amd import synth
    tm_build.build()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, subset = synth.synth_model(1200, seed=1)
    res = synth.resolution(model)
    scene = synth.synth_scene(8000, 1200, seed=2, copies=2, model_resolution=res)
    m = tmo.Model(model, res)
    m.build_voxels(threads=2)
    assert m.build_pairs(subset) == 0
    s = tmo.Scene(scene["surfels"], scene["tangent_mask"])
    pairs = synth.sample_scene_pairs(scene, m.diameter, 16, 4, seed=13)
    mine = tmd.shard_pairs(pairs, rank, world)
    poses, prov = s.generate_hypotheses(m, mine)
    counts, scores, _ = s.score_hypotheses(m, poses, 0.5, 3.0, threads=2)
    if len(poses):
        order = sorted(range(len(poses)), key=lambda i: (-int(counts[i]), -scores[i], i))
        b = order[0]
        rec = tmd.make_record(counts[b], tmd.canonical_hyp_id(int(prov[b][0]), int(prov[b][3]), rank, world),
                              scores[b], poses[b])
    else:
        rec = tmd.empty_record()
    best = tmd.allgather_best(rec)
    pending = tmd.allgather_best_async(rec)  # what bench.py uses: started now, awaited later
    best2 = pending.result()
    assert (best2.inlier_count, best2.hyp_id, best2.score) == (best.inlier_count, best.hyp_id, best.score)
    assert list(best2.pose) == list(best.pose)
    np.save(os.path.join(out_dir, "best_%d.npy" % rank),
            np.array([best.inlier_count, best.hyp_id, best.score] + list(best.pose), dtype=np.float64))
    np.save(os.path.join(out_dir, "local_%d.npy" % rank),
            np.array([rec.inlier_count, rec.hyp_id, rec.score], dtype=np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_best_reduce(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    b0 = np.load(tmp_path / "best_0.npy")
    b1 = np.load(tmp_path / "best_1.npy")
    assert np.array_equal(b0, b1), "ranks disagree on the global best"
    locals_ = [np.load(tmp_path / ("local_%d.npy" % r)) for r in range(2)]
    # the global best is the better of the two local bests: count desc, score desc, id asc
    exp = sorted(locals_, key=lambda r: (-r[0], -r[2], r[1]))[0]
    assert b0[0] == exp[0] and b0[1] == exp[1] and b0[2] == exp[2]
    assert b0[0] > 600  # a planted instance was found by one of the shards
    # ids are canonical: global pair index * 200 + entry, and rank r owns the global pairs r, r + 2, ...
    assert (int(locals_[0][1]) // 200) % 2 == 0 and (int(locals_[1][1]) // 200) % 2 == 1


def test_winner_among_ties_does_not_depend_on_world_size():
    """Hypothesis ids come from provenance (global sample pair, entry), so the tie-break (count desc, score
    desc, id asc) picks the same hypothesis for 1, 2 and 4 ranks. Synthetic table with many exact ties."""
    sys.path.insert(0, ROOT)
    from triplet_match_amd import capi
    from triplet_match_amd import dist as tmd
    rng = np.random.default_rng(3)
    n_pairs = 23
    entries = rng.integers(1, 9, size=n_pairs)  # hypotheses per sample pair
    counts = [rng.integers(5, 8, size=e) for e in entries]  # few distinct values: ties everywhere
    scores = [np.round(rng.random(e), 1) for e in entries]
    winners = []
    for world in (1, 2, 4):
        recs = []
        for rank in range(world):
            best = None
            for lp, p in enumerate(range(rank, n_pairs, world)):  # shard_pairs order
                for e in range(entries[p]):
                    hid = tmd.canonical_hyp_id(lp, e, rank, world)
                    assert hid == p * 200 + e
                    key = (-int(counts[p][e]), -float(scores[p][e]), hid)
                    if best is None or key < best[0]:
                        best = (key, tmd.make_record(counts[p][e], hid, scores[p][e], np.arange(16) + hid))
            recs.append(best[1] if best else tmd.empty_record())
        w = capi.best_select(recs)
        winners.append((w.inlier_count, w.score, w.hyp_id, tuple(w.pose)))
    assert winners[0] == winners[1] == winners[2]


def test_shard_is_a_partition():
    sys.path.insert(0, ROOT)
    from triplet_match_amd import dist as tmd
    pairs = np.arange(202, dtype=np.uint32).reshape(-1, 2)
    for world in (1, 2, 3, 8):
        shards = [tmd.shard_pairs(pairs, r, world) for r in range(world)]
        assert sum(len(s) for s in shards) == len(pairs)
        merged = np.zeros_like(pairs)
        for r, s in enumerate(shards):
            merged[r::world] = s
        assert np.array_equal(merged, pairs)


def _worker_multi(rank, world, port, out_dir, n_models):
    """config 5 wiring: n_models models x sample pairs on a Wm x Wp grid, per-model best reduce"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from oracle import tmo
    from triplet_match_amd import build as tm_build
    from triplet_match_amd import dist as tmd
    from triplet_match_amd import synth
    tm_build.build()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scales = [(1.0, 1.0, 1.0), (0.7, 1.3, 1.1), (1.2, 0.8, 0.9)][:n_models]
    clouds = []
    for k, sc in enumerate(scales):
        M, sub = synth.synth_model(900, seed=1 + k, scale=sc)
        clouds.append((M, sub, synth.resolution(M)))
    # one scene holding an instance of every model
    rng = np.random.default_rng(5)
    parts = []
    for k, (M, sub, res) in enumerate(clouds):
        inst, _ = synth.synth_model(900, seed=50 + k, scale=scales[k])
        R = synth.random_rotation(rng)
        t = np.array([3.0 * k, 1.0, 0.5])
        inst[:, 0:3] = inst[:, 0:3] @ R.T + t
        inst[:, 4:7] = inst[:, 4:7] @ R.T
        inst[:, 9:12] = inst[:, 9:12] @ R.T
        parts.append(inst)
    surf = np.concatenate(parts).astype(np.float32)
    tmask = (np.linalg.norm(surf[:, 9:12], axis=1) > 0.7).astype(np.int32)
    s = tmo.Scene(surf, tmask)
    recs, all_best = [], []
    mine = set(tmd.my_models(rank, world, n_models))
    for k, (M, sub, res) in enumerate(clouds):
        if k not in mine:
            recs.append(tmd.empty_record())
            continue
        m = tmo.Model(M, res)
        m.build_voxels(threads=2)
        assert m.build_pairs(sub) == 0
        pairs = synth.sample_scene_pairs(dict(surfels=surf, tangent_mask=tmask), m.diameter, 12, 3, seed=13)
        shard = tmd.shard_pairs_2d(pairs, rank, world, n_models)
        poses, prov = s.generate_hypotheses(m, shard)
        if len(poses) == 0:
            recs.append(tmd.empty_record())
            continue
        counts, scores, _ = s.score_hypotheses(m, poses, 0.5, 3.0, threads=2)
        b = sorted(range(len(poses)), key=lambda i: (-int(counts[i]), -scores[i], i))[0]
        recs.append(tmd.make_record(counts[b], tmd.canonical_hyp_id_2d(int(prov[b][0]), int(prov[b][3]), rank, world,
                                                                    n_models), scores[b], poses[b]))
    best = tmd.allgather_best_many(recs)
    out = np.array([[b.inlier_count, b.hyp_id, b.score] for b in best], dtype=np.float64)
    loc = np.array([[r.inlier_count, r.hyp_id, r.score] for r in recs], dtype=np.float64)
    np.save(os.path.join(out_dir, "mbest_%d.npy" % rank), out)
    np.save(os.path.join(out_dir, "mlocal_%d.npy" % rank), loc)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_models", [2, 1])
def test_two_rank_gloo_multi_model_reduce(tmp_path, n_models):
    """n_models = 2: grid 2 x 1 (each rank owns one model); n_models = 1: grid 1 x 2 (pairs sharded)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from triplet_match_amd import dist as tmd
    assert tmd.grid_2d(0, 2, n_models)[:2] == ((2, 1) if n_models == 2 else (1, 2))
    assert tmd.grid_2d(5, 8, 32) == (8, 1, 5, 0) and tmd.grid_2d(5, 8, 2) == (2, 4, 1, 1)
    assert sorted(sum((tmd.my_models(r, 8, 32) for r in range(8)), [])) == list(range(32))
    port = _free_port()
    mp.spawn(_worker_multi, args=(2, port, str(tmp_path), n_models), nprocs=2, join=True)
    b0, b1 = np.load(tmp_path / "mbest_0.npy"), np.load(tmp_path / "mbest_1.npy")
    assert np.array_equal(b0, b1) and b0.shape == (n_models, 3)
    l0, l1 = np.load(tmp_path / "mlocal_0.npy"), np.load(tmp_path / "mlocal_1.npy")
    for m in range(n_models):
        exp = sorted([l0[m], l1[m]], key=lambda r: (-r[0], -r[2], r[1]))[0]
        assert np.array_equal(b0[m], exp)
        assert b0[m][0] > 400  # every model finds its own instance


# ---- bench.py's own launcher (`python bench.py --gpus N` without torchrun around it) ----
def _run_bench(argv, env_extra=None, drop=("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=600)


def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher in the environment: the parent starts the two ranks before
    anything touches a GPU, rank 0's single JSON line comes through, n_gpus is 2. (--launcher-selftest: the ranks
    exchange a synthetic record over gloo instead of running kernels — there is no GPU here.)"""
    import json
    r = _run_bench(["--gpus", "2", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["comm_world"] == 2 and out["selftest"] is True
    assert out["best"] == {"inlier_count": 101, "hyp_id": 8}  # rank 1's record wins on the count
    assert out["value"] is None  # a self-test never carries a figure


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus 8 under a launcher that started one rank (or the reverse) must not print a 1-GPU line."""
    r = _run_bench(["--gpus", "8", "--launcher-selftest"],
                   env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode == 2 and r.stdout.strip() == "", (r.returncode, r.stdout)
    assert "--gpus 8" in r.stderr
    r = _run_bench(["--gpus", "1", "--launcher-selftest"],
                   env_extra={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode == 2 and r.stdout.strip() == ""
