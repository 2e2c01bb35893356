"""Multi-GPU composition of the hot path: the haystack list is sharded by contiguous index range, one process
per GPU scores its shard with a global `index_offset` (exactly what `match_list_parallel`'s workers do with
2048-item chunks, reference src/matcher/parallel.rs:55-63), then the per-shard, index-ordered match lists are
exchanged with ONE all-gather (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests) and combined on
the host the way the reference combines its per-thread runs: reverse / radix sort per run, then k-way merge
(parallel.rs:66-87).  Scoring itself needs no collective."""
import numpy as np
import torch
import torch.distributed as dist

from . import MATCH_DTYPE, SortStrategy, k_merge_matches, radix_sort_matches


def shard_range(n_total, rank, world):
    """Contiguous index range [lo, hi) of shard `rank` (g*ceil(N/G) .. min((g+1)*ceil(N/G), N))."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def all_gather_matches(records_u8, count, group=None):
    """records_u8: uint8 tensor holding >= count 8-byte records (device or CPU); count: python int or 0-dim/1-elem int tensor.
    Returns a list (one per rank) of numpy MATCH_DTYPE arrays.  Two collectives: counts, then records padded to the max count
    (all-gather-v emulation)."""
    world = dist.get_world_size(group)
    dev = records_u8.device
    cnt = count.to(torch.int64).reshape(1) if torch.is_tensor(count) else torch.tensor([int(count)], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, cnt.to(dev), group=group)
    counts_h = counts.cpu().tolist()
    mx = max(max(counts_h), 1)
    send = records_u8.reshape(-1)[: mx * 8]
    if send.numel() < mx * 8:  # local buffer shorter than the largest shard's result: pad
        pad = torch.zeros(mx * 8, dtype=torch.uint8, device=dev)
        pad[: send.numel()] = send
        send = pad
    recv = torch.empty(world * mx * 8, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    host = recv.cpu().numpy().reshape(world, mx * 8)
    return [host[r, : counts_h[r] * 8].copy().view(MATCH_DTYPE) for r in range(world)]


def merge_shard_runs(runs, sort):
    """Per-shard index-ordered runs -> the reference's final ordering (parallel.rs:66-87)."""
    sort = SortStrategy(int(sort))
    prepared = []
    for r in runs:
        r = np.ascontiguousarray(r)
        if sort in (SortStrategy.IndexDesc, SortStrategy.ScoreThenIndexDesc):
            r = r[::-1].copy()
        if sort in (SortStrategy.ScoreThenIndexAsc, SortStrategy.ScoreThenIndexDesc):
            r = radix_sort_matches(r)
        prepared.append(r)
    return k_merge_matches(sort, prepared)
