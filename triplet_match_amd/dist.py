"""Multi-GPU plumbing of the hot path: hypothesis sharding and the best-record reduction.

The path shards embarrassingly: scene, model, voxel grid and pair table are replicated,
scene sample pairs are dealt round-robin to the ranks, every rank generates and scores its own
hypotheses, and ONE small collective per batch exchanges the 96-byte best records (all-gather over
torch.distributed: backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests). The argmax
itself is tm_best_select (C ABI): count desc, score desc, hypothesis id asc — deterministic and
independent of the number of ranks because a hypothesis' id comes from its provenance, not from where it
was scored: id = global sample pair index * query_limit + entry within the pair (canonical_hyp_id).

On the GPUs the exchange itself is the C ABI's tm_best_reduce (one ncclAllGather of device-resident
records on the library's stream, capi.Comm); the torch.distributed functions below carry the same
records over gloo in the CPU tests and in the one-GPU rehearsal of bench.py.
"""
import ctypes as C

import numpy as np

from . import capi

RECORD_BYTES = 96


def shard_pairs(pairs, rank, world):
    """Round-robin shard of the ordered scene sample pairs (global pair p -> rank p % world)."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    return np.ascontiguousarray(pairs[rank::world])


QUERY_LIMIT = 200  # include/impl/scene.hpp:19


def global_pair_index(local_pair, rank, world):
    """shard_pairs deals global pair p to rank p % world at local position p // world."""
    return rank + local_pair * world


def canonical_hyp_id(local_pair, entry, rank, world, query_limit=QUERY_LIMIT):
    """Id of the hypothesis built from entry `entry` (visit order inside its sample pair) of this rank's
    local pair `local_pair`: the same number whatever the number of ranks (the provenance rows written by
    generate_hypotheses are (local pair, m_i, m_j, entry))."""
    return global_pair_index(local_pair, rank, world) * query_limit + entry


def record_to_bytes(rec):
    return np.frombuffer(bytes(rec), dtype=np.uint8).copy()


def allgather_best(rec, device=None, group=None):
    """All-gather the per-rank best records and select the global best. Every rank returns the same
    BestRecord. `device` = torch device for the staging tensors (cuda for nccl, cpu for gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return capi.best_select([rec])
    local = torch.from_numpy(record_to_bytes(rec))
    if device is not None:
        local = local.to(device)
    gathered = torch.empty(RECORD_BYTES * world, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    raw = gathered.cpu().numpy().tobytes()
    recs = [capi.BestRecord.from_buffer_copy(raw[RECORD_BYTES * r:RECORD_BYTES * (r + 1)]) for r in range(world)]
    return capi.best_select(recs)


class PendingBest:
    """An all-gather of best records in flight (allgather_best_async); result() waits and selects."""

    def __init__(self, work, gathered, world, local):
        self._work, self._gathered, self._world, self._local = work, gathered, world, local

    def result(self):
        if self._work is None:
            return capi.best_select([self._local])
        self._work.wait()
        raw = self._gathered.cpu().numpy().tobytes()
        recs = [capi.BestRecord.from_buffer_copy(raw[RECORD_BYTES * r:RECORD_BYTES * (r + 1)])
                for r in range(self._world)]
        return capi.best_select(recs)


def allgather_best_async(rec, device=None, group=None):
    """Starts the 96-byte-per-rank all-gather and returns at once, so that the next batch's kernels (on the
    library's own stream) overlap the exchange; call .result() when the global best is needed."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return PendingBest(None, None, 1, rec)
    local = torch.from_numpy(record_to_bytes(rec))
    if device is not None:
        local = local.to(device, non_blocking=True)
    gathered = torch.empty(RECORD_BYTES * world, dtype=torch.uint8, device=local.device)
    work = dist.all_gather_into_tensor(gathered, local, group=group, async_op=True)
    return PendingBest(work, gathered, world, rec)


def make_record(count, hyp_id, score, pose16):
    r = capi.BestRecord()
    r.inlier_count = int(count)
    r.hyp_id = int(hyp_id)
    r.score = float(score)
    for k in range(16):
        r.pose[k] = float(pose16[k])
    return r


def empty_record():
    r = capi.BestRecord()
    r.hyp_id = 0xFFFFFFFF
    return r


# ---- multi-model search (BASELINE config 5): models x sample pairs sharded 2-D, per-model reduce ----
def grid_2d(rank, world, n_models):
    """Splits `world` ranks into Wm model groups x Wp pair shards (Wm = the largest divisor of world that
    is <= n_models). Returns (Wm, Wp, model_group, pair_shard): this rank searches the models
    m % Wm == model_group with the sample pairs p % Wp == pair_shard of each."""
    wm = max(d for d in range(1, world + 1) if world % d == 0 and d <= max(1, n_models))
    wp = world // wm
    return wm, wp, rank // wp, rank % wp


def my_models(rank, world, n_models):
    wm, _, g, _ = grid_2d(rank, world, n_models)
    return list(range(g, n_models, wm))


def shard_pairs_2d(pairs, rank, world, n_models):
    """The pair shard of this rank inside its model group (global pair p -> shard p % Wp)."""
    _, wp, _, shard = grid_2d(rank, world, n_models)
    pairs = np.asarray(pairs).reshape(-1, 2)
    return np.ascontiguousarray(pairs[shard::wp])


def canonical_hyp_id_2d(local_pair, entry, rank, world, n_models, query_limit=QUERY_LIMIT):
    """canonical_hyp_id inside this rank's model group (pairs are sharded over its Wp pair shards)."""
    _, wp, _, shard = grid_2d(rank, world, n_models)
    return (shard + local_pair * wp) * query_limit + entry


def allgather_best_many(recs, device=None, group=None):
    """Per-model best records. `recs[m]` is this rank's best for model m (empty_record() for models it
    did not search). ONE all-gather of world x n_models x 96 bytes, then tm_best_select per model;
    every rank returns the same list."""
    import torch
    import torch.distributed as dist
    n_models = len(recs)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [capi.best_select([r]) for r in recs]
    local = torch.from_numpy(np.concatenate([record_to_bytes(r) for r in recs]))
    if device is not None:
        local = local.to(device)
    gathered = torch.empty(RECORD_BYTES * n_models * world, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    raw = gathered.cpu().numpy().tobytes()
    out = []
    for m in range(n_models):
        cand = []
        for r in range(world):
            o = RECORD_BYTES * (r * n_models + m)
            cand.append(capi.BestRecord.from_buffer_copy(raw[o:o + RECORD_BYTES]))
        out.append(capi.best_select(cand))
    return out
