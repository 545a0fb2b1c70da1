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


def shard_plan(lengths: Sequence[int], world_size: int) -> np.ndarray:
    """rank_of_song as computed by the C ABI (blissgpu_shard_plan: device-free, the plan blissgpu_node_* uses); the
    same assignment as shard_songs, which is the readable statement of the rule."""
    import ctypes as C

    from . import _ffi

    lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
    ranks = np.empty(len(lengths), np.uint32)
    _ffi.check(_ffi.lib().blissgpu_shard_plan(lengths.ctypes.data_as(C.POINTER(C.c_uint64)), len(lengths), world_size,
                                              ranks.ctypes.data_as(C.POINTER(C.c_uint32))))
    return ranks


def all_gather_features(local_rows, local_indices, n_total: int, group=None, n_local_max=None):
    """All-gather ragged [n_local, d] feature blocks and scatter them to their global rows.

    local_rows: torch tensor [n_local, d] (CUDA -> RCCL, CPU -> gloo); local_indices: global song index
    of each local row.  Returns a [n_total, d] tensor on the same device, identical on every rank.
    n_local_max: the largest shard size if the caller knows it (e.g. from shard_songs, which every rank computes
    identically); it saves the count exchange and its host synchronisation.

    ONE collective per call: the global row index travels as an extra column of the row itself (song counts stay far
    below 2^24, so an f32 column holds them exactly), the blocks are padded to the largest shard (<= ~1 MB per rank at
    library sizes: latency-bound on xGMI), and one index_copy scatters all ranks' rows at once -- no host
    synchronisation, no per-rank loop."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_local, d = local_rows.shape
    dev = local_rows.device
    assert n_total < (1 << 24), "row indices travel as f32"
    idx = local_indices.to(dev) if torch.is_tensor(local_indices) else torch.as_tensor(np.asarray(local_indices, dtype=np.int64), device=dev)
    if n_local_max is None:
        counts = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(counts, torch.tensor([n_local], dtype=torch.int64, device=dev), group=group)
        n_max = int(counts.max().item())
    else:
        n_max = int(n_local_max)
        assert n_local <= n_max
    pack = torch.zeros((n_max, d + 1), dtype=local_rows.dtype, device=dev)
    pack[:, d] = -1.0                                  # padding rows
    pack[:n_local, :d] = local_rows
    pack[:n_local, d] = idx.to(local_rows.dtype)
    gathered = torch.empty((world * n_max, d + 1), dtype=local_rows.dtype, device=dev)
    dist.all_gather_into_tensor(gathered, pack, group=group)
    rows = gathered[:, d].to(torch.int64)
    rows = torch.where(rows < 0, torch.full_like(rows, n_total), rows)   # padding lands in one spare row past the end
    full = torch.full((n_total + 1, d), float("nan"), dtype=local_rows.dtype, device=dev)
    full.index_copy_(0, rows, gathered[:, :d])
    return full[:n_total]


def row_block(n_rows: int, rank: int, world_size: int):
    """Row range [lo, hi) of the pairwise-distance matrix computed by `rank` (blissgpu_row_block is the C form)."""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
