"""Multi-GPU SSV: one process per GPU, the sequence cut into runs of whole 12288-column segments.

A cell depends only on its own diagonal, so rank r sweeps
the diagonals that reach its columns from their start (a left halo of nrows-1
columns is recomputed: C4, sum L = 1e6 over 1.25e8 columns per rank, 0.4 % extra
work) and reports only the hits inside its columns.  No data-path collective.
Each rank orders its own records; because the shards are runs of whole segments,
the lists concatenated in rank order ARE the reference's device order, so the
only exchange is the gather of the records to rank 0 (RCCL over xGMI when the
backend is "nccl"; the same code runs over gloo on CPU tensors in the tests) and
rank 0 never sorts.  The reference has no multi-device path at all
(host/Havac.hpp:51: one deviceIndex per object).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def gather_hits(local_hits: torch.Tensor, local_count: int, group=None):
    """All ranks call this with their own records (int64 view of the packed u64, first
    `local_count` valid).  Returns (records concatenated in rank order, per-rank counts) on
    rank 0 and (None, counts) elsewhere.  Two collectives: an all_gather of the counts
    (8 B per rank), then a gather to rank 0 of the records padded to the largest count
    (payload is KB..MB per rank)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out_dev = local_hits.device
    # gloo (CPU rehearsals and the tests) moves host tensors; nccl = RCCL moves device tensors over xGMI
    dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else out_dev
    capacity = local_hits.numel()
    local_hits = local_hits.to(dev) if dev != out_dev else local_hits
    mine = torch.tensor([local_count], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, mine, group=group)
    counts = torch.cat(counts).tolist()          # one device-to-host sync for all ranks' counts
    biggest = max(counts)
    if biggest == 0:
        return (local_hits[:0].to(out_dev) if rank == 0 else None), counts
    if biggest > capacity:
        raise ValueError("ranks must use hit buffers of one capacity (a rank reported more records than fit here)")
    padded = local_hits[:biggest]                  # no copy: records beyond local_count are ignored by rank 0
    parts = [torch.empty(biggest, dtype=local_hits.dtype, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, parts, dst=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if rank != 0:
        return None, counts
    return torch.cat([p[:c] for p, c in zip(parts, counts)]).to(out_dev), counts


class ShardedSsv:
    """Rank-local driver: enqueue this rank's shard, order its hits, gather to rank 0 (already in order)."""

    def __init__(self, hit_capacity: int, device: torch.device):
        from .ssv import SsvContext
        self.ctx = SsvContext()
        self.device = device
        self.hits = torch.empty(hit_capacity, dtype=torch.int64, device=device)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def run(self, d_seq: torch.Tensor, nsymbols: int, d_phmm: torch.Tensor, nrows: int):
        """-> (records on rank 0 in device order or None, hits found by this rank)"""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx.enqueue(d_seq.data_ptr(), nsymbols, d_phmm.data_ptr(), nrows, self.hits.data_ptr(),
                         self.hits.numel(), self.rank, self.world, 0, stream)
        found = self.ctx.finish()
        if self.world == 1:
            return self.hits[:found], found
        merged, _ = gather_hits(self.hits, found)
        return merged, found
