"""Multi-GPU plumbing of the hot path: hypothesis sharding and the best-record reduction.

The path shards embarrassingly: scene, model, voxel grid and pair table are replicated,
scene sample pairs are dealt round-robin to the ranks, every rank generates and scores its own
hypotheses, and ONE small collective per batch exchanges the 96-byte best records (all-gather over
torch.distributed: backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests). The argmax
itself is tm_best_select (C ABI): count desc, score desc, hypothesis id asc — deterministic and
independent of the number of ranks because ids are global (rank + k * world).
"""
import ctypes as C

import numpy as np

from . import capi

RECORD_BYTES = 96


def shard_pairs(pairs, rank, world):
    """Round-robin shard of the ordered scene sample pairs (global pair p -> rank p % world)."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    return np.ascontiguousarray(pairs[rank::world])


def global_hyp_id(local_index, rank, world):
    return rank + local_index * world


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
