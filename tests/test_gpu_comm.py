"""GPU tests of the multi-GPU entry points of the C ABI with a communicator of ONE rank (the box has one GPU;
RCCL refuses two ranks per device): tm_comm_create runs ncclCommInitRank, tm_best_reduce runs ncclAllGather on
the library's stream + the selection kernel, exactly the calls N ranks make. The N > 1 logic is covered on the
CPU (tests/test_distributed.py, tests/test_cpp_dropin.py::test_rank_sharded_search_equals_single_process)."""
import ctypes as C

import numpy as np
import pytest

from triplet_match_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scored(gpu_ctx):
    model, subset = synth.synth_model(3000, seed=1)
    res = synth.resolution(model)
    scene = synth.synth_scene(20000, 3000, seed=2, copies=2, model_resolution=res)
    gm = capi.Model(gpu_ctx, model, res)
    gm.build_voxels()
    gm.build_pairs(subset)
    gs = capi.Scene(gpu_ctx, scene["surfels"], scene["tangent_mask"])
    pairs = synth.sample_scene_pairs(scene, gm.diameter, 24, 6, seed=13)
    cap = 1 << 14
    d_ij = gpu_ctx.alloc(pairs.nbytes)
    gpu_ctx.h2d(d_ij, pairs)
    d_poses, d_prov = gpu_ctx.alloc(cap * 64), gpu_ctx.alloc(cap * 16)
    d_counts, d_scores = gpu_ctx.alloc(cap * 4), gpu_ctx.alloc(cap * 8)
    n, _ = gs.generate_hypotheses_dev(gm, d_ij, len(pairs), d_poses, d_prov, cap)
    assert n > 100
    gs.score_hypotheses_dev(gm, d_poses, n, 3.0, d_counts, d_scores)
    counts, scores = np.zeros(n, np.uint32), np.zeros(n, np.float64)
    prov, poses = np.zeros((n, 4), np.uint32), np.zeros((n, 16), np.float32)
    gpu_ctx.d2h(counts, d_counts)
    gpu_ctx.d2h(scores, d_scores)
    gpu_ctx.d2h(prov, d_prov)
    gpu_ctx.d2h(poses, d_poses)
    return dict(n=n, d=(d_counts, d_scores, d_poses, d_prov), counts=counts, scores=scores, prov=prov, poses=poses,
                keep=(gm, gs))


def _expected(scored, pair_offset, pair_stride, limit=200):
    ids = (pair_offset + scored["prov"][:, 0].astype(np.int64) * pair_stride) * limit + scored["prov"][:, 3]
    order = sorted(range(scored["n"]), key=lambda i: (-int(scored["counts"][i]), -scored["scores"][i], int(ids[i])))
    b = order[0]
    return int(scored["counts"][b]), int(ids[b]), float(scored["scores"][b]), scored["poses"][b]


@pytest.mark.parametrize("with_id", [True, False])
def test_best_reduce_single_rank(gpu_ctx, scored, with_id):
    uid = capi.comm_unique_id() if with_id else None
    comm = capi.Comm(gpu_ctx, 0, 1, uid)
    d_counts, d_scores, d_poses, d_prov = scored["d"]
    d_best = gpu_ctx.alloc(96 * 2)
    for off, stride in ((0, 1), (3, 8)):
        gpu_ctx.best_of_batch_prov(d_counts, d_scores, d_poses, d_prov, scored["n"], d_best, pair_offset=off,
                                   pair_stride=stride)
        best = comm.best_reduce(d_best, 1)[0]
        cnt, hid, sc, pose = _expected(scored, off, stride)
        assert (best.inlier_count, best.hyp_id, best.score) == (cnt, hid, sc)
        assert np.array_equal(np.array(list(best.pose), np.float32).view(np.uint32), pose.view(np.uint32))
    comm.close()


def test_best_reduce_per_model_and_empty_records(gpu_ctx, scored):
    """two 'models': record 0 = a real local best, record 1 = empty (a model this rank did not search)"""
    comm = capi.Comm(gpu_ctx, 0, 1, capi.comm_unique_id())
    d_counts, d_scores, d_poses, d_prov = scored["d"]
    d_best = gpu_ctx.alloc(96 * 2)
    gpu_ctx.best_of_batch_prov(d_counts, d_scores, d_poses, d_prov, scored["n"], d_best)
    empty = capi.BestRecord()
    empty.hyp_id = 0xFFFFFFFF
    gpu_ctx.h2d(C.c_void_p(d_best.value + 96), np.frombuffer(bytes(empty), np.uint8))
    out = comm.best_reduce(d_best, 2)
    assert out[0].inlier_count == _expected(scored, 0, 1)[0] and out[1].hyp_id == 0xFFFFFFFF
    assert out[1].inlier_count == 0
    # a batch without hypotheses gives the empty record
    gpu_ctx.best_of_batch_prov(d_counts, d_scores, d_poses, d_prov, 0, d_best)
    assert comm.best_reduce(d_best, 1)[0].hyp_id == 0xFFFFFFFF


def test_allgather_records_single_rank(gpu_ctx):
    comm = capi.Comm(gpu_ctx, 0, 1, capi.comm_unique_id())
    recs = []
    for k in range(3):
        r = capi.BestRecord()
        r.inlier_count, r.hyp_id, r.score = 10 + k, k, 0.5 * k
        r.pose[5] = float(k)
        recs.append(r)
    out = comm.allgather_records(recs)
    assert [bytes(a) for a in out] == [bytes(b) for b in recs]


@pytest.mark.parametrize("route", ["no_comm", "one_rank_rccl"])
def test_best_reduce_in_two_halves_overlaps_and_is_bounded(gpu_ctx, scored, route):
    """tm_best_reduce_begin / _end: begin returns without waiting, several reductions may be in flight and are
    handed out in any order, the fifth begin without an end is refused (TM_ERR_CAPACITY), a ticket is good once."""
    comm = capi.Comm(gpu_ctx, 0, 1, capi.comm_unique_id()) if route == "one_rank_rccl" else None
    d_counts, d_scores, d_poses, d_prov = scored["d"]
    d_best = [gpu_ctx.alloc(96) for _ in range(4)]
    params = [(0, 1), (3, 8), (1, 2), (5, 7)]
    pend = []
    for d, (off, stride) in zip(d_best, params):
        gpu_ctx.best_of_batch_prov(d_counts, d_scores, d_poses, d_prov, scored["n"], d, pair_offset=off,
                                   pair_stride=stride)
        pend.append(capi.best_reduce_begin(gpu_ctx, comm, d, 1))
    with pytest.raises(capi.TmError) as e:
        capi.best_reduce_begin(gpu_ctx, comm, d_best[0], 1)
    assert e.value.code == capi.ERR_CAPACITY
    for k in (2, 0, 3, 1):
        best = pend[k].result()[0]
        cnt, hid, sc, _ = _expected(scored, *params[k])
        assert (best.inlier_count, best.hyp_id, best.score) == (cnt, hid, sc)
    out = (capi.BestRecord * 1)()
    assert capi.lib().tm_best_reduce_end(gpu_ctx.h, pend[0].ticket, out) == capi.ERR_INVALID
    # the slots are free again
    assert capi.best_reduce_begin(gpu_ctx, comm, d_best[0], 1).result()[0].inlier_count == _expected(scored, 0, 1)[0]
    # a stale ticket stays refused after its slot has been handed out again and is busy (the ticket carries the
    # slot's generation), and the other caller's reduction is untouched
    stale = [p.ticket for p in pend]
    again = [capi.best_reduce_begin(gpu_ctx, comm, d_best[k], 1) for k in range(4)]
    assert not set(stale) & {p.ticket for p in again}
    for t in stale:
        assert capi.lib().tm_best_reduce_end(gpu_ctx.h, t, out) == capi.ERR_INVALID
    # tickets ended out of order leave holes: a new begin takes any free slot instead of CAPACITY
    assert again[2].result()[0].inlier_count == _expected(scored, *params[2])[0]
    assert again[0].result()[0].inlier_count == _expected(scored, *params[0])[0]
    more = [capi.best_reduce_begin(gpu_ctx, comm, d_best[1], 1), capi.best_reduce_begin(gpu_ctx, comm, d_best[3], 1)]
    with pytest.raises(capi.TmError) as e:
        capi.best_reduce_begin(gpu_ctx, comm, d_best[0], 1)
    assert e.value.code == capi.ERR_CAPACITY
    for p, k in ((again[1], 1), (again[3], 3), (more[0], 1), (more[1], 3)):
        assert p.result()[0].inlier_count == _expected(scored, *params[k])[0]
    assert capi.comm_probe() is True
    if comm is not None:
        comm.close()


def test_library_communicator_before_torch_exits_cleanly():
    """The order that once ended in `double free or corruption (!prev)` at interpreter exit (round 2,
    gpurun_out/r02_gputests_full{,2,3,4}.log): the library dlopens RCCL first, torch (which brings its own copy)
    is imported afterwards. librccl is now loaded RTLD_LOCAL (csrc/tm_comm.hip), so nothing interposes: the child
    must leave with exit code 0, not a signal."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from triplet_match_amd import capi\n"
            "ctx = capi.Context(0)\n"
            "comm = capi.Comm(ctx, 0, 1, capi.comm_unique_id())\n"
            "assert 'torch' not in sys.modules\n"
            "import torch\n"
            "x = torch.ones(8, device='cuda').sum().item()\n"
            "assert x == 8.0\n"
            "comm.close()\n"
            "print('ok')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    assert r.stdout.strip().endswith("ok")
