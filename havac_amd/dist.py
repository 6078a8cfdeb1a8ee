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


class _Slot:
    def __init__(self, hit_capacity, device, own_stream):
        from .ssv import SsvContext
        self.ctx = SsvContext()
        self.hits = torch.empty(hit_capacity, dtype=torch.int64, device=device)
        self.stream = torch.cuda.Stream(device) if own_stream else None
        self.kernel_done = None          # recorded behind the SSV kernel of the pass in flight


class ShardedSsv:
    """Rank-local driver: enqueue this rank's shard, order its hits, gather to rank 0 (already in order).

    depth > 1 keeps that many passes in flight, each with its own context, hit buffer and HIP stream:
    ``submit`` enqueues a pass, ``collect`` finishes the oldest one.  While the host waits for pass k's hit count,
    orders its records and (N > 1) gathers them over RCCL, the SSV kernel of pass k+1 is already running.  The SSV
    kernels themselves are kept back to back, never side by side (pass k+1 waits for the kernel of pass k), so a
    kernel's event-timed duration stays the duration of that kernel alone."""

    def __init__(self, hit_capacity: int, device: torch.device, depth: int = 1, back_to_back: bool = True,
                 gather_when_alone: bool = False):
        self.device = device
        self.back_to_back = back_to_back
        # rehearsals on one GPU: run the collectives even in a one-rank group
        self.gather_when_alone = gather_when_alone and dist.is_initialized()
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.slots = [_Slot(hit_capacity, device, depth > 1) for _ in range(max(1, depth))]
        self.in_flight = []               # slot indices, oldest first
        self.next_slot = 0
        self.last_kernel_done = None
        self.ctx = self.slots[0].ctx      # the context of the most recently collected pass (for last_ms)
        self.hits = self.slots[0].hits

    def submit(self, d_seq: torch.Tensor, nsymbols: int, d_phmm: torch.Tensor, nrows: int):
        if len(self.in_flight) == len(self.slots):
            raise RuntimeError("every slot is in flight: collect() first")
        slot = self.slots[self.next_slot]
        stream = slot.stream if slot.stream is not None else torch.cuda.current_stream(self.device)
        if slot.stream is not None:
            stream.wait_stream(torch.cuda.current_stream(self.device))      # the caller's inputs
            if self.back_to_back and self.last_kernel_done is not None:
                stream.wait_event(self.last_kernel_done)
        slot.ctx.enqueue(d_seq.data_ptr(), nsymbols, d_phmm.data_ptr(), nrows, slot.hits.data_ptr(),
                         slot.hits.numel(), self.rank, self.world, 0, stream.cuda_stream)
        if slot.stream is not None:
            slot.kernel_done = torch.cuda.Event()
            slot.kernel_done.record(stream)
            self.last_kernel_done = slot.kernel_done
        self.in_flight.append(self.next_slot)
        self.next_slot = (self.next_slot + 1) % len(self.slots)

    def collect(self):
        """-> (records on rank 0 in device order or None, hits found by this rank) of the oldest pass in flight"""
        slot = self.slots[self.in_flight.pop(0)]
        found = slot.ctx.finish()
        self.ctx, self.hits = slot.ctx, slot.hits
        if self.world == 1 and not self.gather_when_alone:
            return slot.hits[:found], found
        if slot.stream is None:
            merged, _ = gather_hits(slot.hits, found)
        else:
            with torch.cuda.stream(slot.stream):
                merged, _ = gather_hits(slot.hits, found)
        return merged, found

    def run(self, d_seq: torch.Tensor, nsymbols: int, d_phmm: torch.Tensor, nrows: int):
        """one pass, start to end -> (records on rank 0 in device order or None, hits found by this rank)"""
        self.submit(d_seq, nsymbols, d_phmm, nrows)
        return self.collect()

    def close(self):
        for slot in self.slots:
            slot.ctx.close()
