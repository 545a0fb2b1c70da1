"""Sharding of a song library across the GPUs of one node.

Songs are independent (no cross-song state anywhere in Song::analyze; the reference's bulk path
already treats them so, src/song/decoder.rs:300-328), so the path shards by song with no data-path
collective.  The only exchange is one all-gather of the [n_local, d] feature rows (RCCL over xGMI
when the tensors live on GPUs, gloo on CPU in the tests) so that every rank holds the full feature
matrix for the pairwise-distance kernel, which is then row-block sharded.
"""
from typing import List, Sequence

import numpy as np


def shard_songs(lengths: Sequence[int], world_size: int) -> List[np.ndarray]:
    """Greedy longest-first assignment balancing the total number of samples per rank (the GPU work
    is proportional to samples).  Returns, per rank, the sorted indices of its songs."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(world_size, np.int64)
    count = np.zeros(world_size, np.int64)
    shards = [[] for _ in range(world_size)]
    for i in order:
        # least loaded rank; ties -> fewest songs -> lowest rank (deterministic on every rank)
        r = int(np.lexsort((np.arange(world_size), count, load))[0])
        shards[r].append(int(i))
        load[r] += lengths[i]
        count[r] += 1
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


def all_gather_features(local_rows, local_indices, n_total: int, group=None, n_local_max=None):
    """All-gather ragged [n_local, d] feature blocks and scatter them to their global rows.

    local_rows: torch tensor [n_local, d] (CUDA -> RCCL, CPU -> gloo); local_indices: global song index
    of each local row.  Returns a [n_total, d] tensor on the same device, identical on every rank.
    n_local_max: the largest shard size if the caller knows it (e.g. from shard_songs, which every rank computes
    identically); it saves the count exchange and its host synchronisation."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    d = local_rows.shape[1]
    dev = local_rows.device
    idx = local_indices.to(dev) if torch.is_tensor(local_indices) else torch.as_tensor(np.asarray(local_indices, dtype=np.int64), device=dev)
    if n_local_max is None:
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=dev), group=group)
        n_max = int(max(int(c.item()) for c in counts))
    else:
        n_max = int(n_local_max)
        assert local_rows.shape[0] <= n_max
    # pad to the largest shard: one fixed-size all-gather (latency-bound at these sizes: <= ~1 MB)
    pad_rows = torch.zeros((n_max, d), dtype=local_rows.dtype, device=dev)
    pad_rows[: local_rows.shape[0]] = local_rows
    pad_idx = torch.full((n_max,), -1, dtype=torch.int64, device=dev)
    pad_idx[: idx.shape[0]] = idx
    rows = [torch.empty_like(pad_rows) for _ in range(world)]
    idxs = [torch.empty_like(pad_idx) for _ in range(world)]
    dist.all_gather(rows, pad_rows, group=group)
    dist.all_gather(idxs, pad_idx, group=group)
    # scatter without a host synchronisation: padding rows (index -1) land in one spare row past the end
    full = torch.full((n_total + 1, d), float("nan"), dtype=local_rows.dtype, device=dev)
    for r, i in zip(rows, idxs):
        full.index_copy_(0, torch.where(i < 0, torch.full_like(i, n_total), i), r)
    return full[:n_total]


def row_block(n_rows: int, rank: int, world_size: int):
    """Row range [lo, hi) of the pairwise-distance matrix computed by `rank`."""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
