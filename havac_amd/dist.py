"""Multi-GPU SSV: one process per GPU, the sequence cut into runs of whole 12288-column segments.

A cell depends only on its own diagonal, so rank r sweeps
the diagonals that reach its columns from their start (a left halo of nrows-1
columns is recomputed: C4, sum L = 5e5 over 1.25e8 columns per rank, 0.2 % extra
work) and reports only the hits inside its columns.  No data-path collective.
Each rank orders its own records; because the shards are runs of whole segments,
the lists concatenated in rank order ARE the reference's device order, so the
only exchange is the gather of the records to rank 0 (RCCL over xGMI when the
backend is "nccl"; the same code runs over gloo on CPU tensors in the tests) and
rank 0 never sorts.  The reference has no multi-device path at all
(host/Havac.hpp:51: one deviceIndex per object).

On the nccl backend the gather is libhavac_dev.so's own (include/havac_dev.h level 3, havac_gather_*: RCCL called behind
the C ABI; torch.distributed only carries the communicator's 128-byte id); over gloo -- the CPU tests, rehearsals of
several ranks on one GPU -- the same exchange goes through torch.distributed's point-to-point operations.

The gather is variable-length: after one all_gather of the per-rank counts
(8 B per rank) every rank r > 0 sends exactly its `count[r]` records and rank 0
receives each list straight into ONE buffer of sum(count) records at the
exclusive-scan offset of its rank (grouped point-to-point: ncclSend/ncclRecv
inside one ncclGroup over RCCL).  Nothing is padded to the largest list and
nothing is concatenated afterwards, so rank 0's extra memory is sum(count) * 8 B
(C4: 4.5e9 records = 36 GB; the padded gather + torch.cat of round 1 needed
more than twice that).  A rank whose pass failed (hit-buffer overflow, HIP
error) reports the count -1: the collective still runs on every rank and every
rank raises afterwards, so no rank is left waiting inside a collective.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

FAILED = -1      # count reported by a rank whose pass raised

# How the records travel (arguments of the caller, not environment variables):
#   route "c_abi" -- libhavac_dev.so's own RCCL calls (include/havac_dev.h level 3): the default on the nccl backend, and on any
#                    backend once use_gather_library() has named a stand-in for RCCL (rehearsals of several ranks on one GPU);
#   route "torch" -- torch.distributed's point-to-point operations: gloo on CPU tensors in the tests, or NCCL for comparison.
_route = "c_abi"
_library = None          # path handed to havac_gather_use_library, or None = librccl.so.1
_deadline_ms = 0         # havac_gather_set_deadline of every communicator made from here on (0: none)


def set_gather_route(route: str):
    global _route
    if route not in ("c_abi", "torch"):
        raise ValueError("route: 'c_abi' or 'torch'")
    _route = route


def use_gather_library(path):
    """Bind `path` instead of librccl.so.1 for the C-ABI gather (before the process's first gather; tests/native/rccl_standin.cpp
    for several ranks on one GPU).  None = RCCL."""
    global _library
    from . import _lib
    rc = _lib.load().havac_gather_use_library(path.encode() if path else None)
    if rc != 0:
        raise RuntimeError("a collective library is bound already: havac_gather_use_library must come before the first gather")
    _library = path or None


def set_gather_deadline(seconds: float):
    """Every host wait of the C-ABI gather gives up after `seconds` (0: never) with hw_client.CollectiveTimeout naming the rank
    and the stage, instead of holding the rank inside a collective a peer never reaches."""
    global _deadline_ms
    _deadline_ms = int(max(0.0, seconds) * 1000)
    for g in _c_gathers.values():
        if g is not None:
            g.set_deadline(_deadline_ms)


def _c_route(group) -> bool:
    return _route == "c_abi" and (dist.get_backend(group) == "nccl" or _library is not None)


class RcclGather:
    """The gather behind the C ABI (include/havac_dev.h, level 3: havac_gather_*): RCCL called from libhavac_dev.so, no torch
    type and no torch kernel anywhere near the records (C4's gathered list is 4.46e9 of them, beyond the 2^32 elements up to
    which this torch build's kernels index correctly).  torch.distributed is only the side channel for the 128-byte id."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        from . import _lib
        self._L = _lib.load()
        self.rank, self.world = rank, world
        h = C.c_void_p()
        rc = self._L.havac_gather_create(rank, world, C.c_char_p(unique_id), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"havac_gather_create failed ({rc}): RCCL could not be loaded or the communicator could not be made")
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        from . import _lib
        buf = C.create_string_buffer(128)
        rc = _lib.load().havac_gather_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"havac_gather_unique_id failed ({rc}): librccl.so.1 could not be loaded")
        return buf.raw

    @staticmethod
    def rccl_version() -> int:
        from . import _lib
        v = C.c_int(0)
        rc = _lib.load().havac_gather_rccl_version(C.byref(v))
        return v.value if rc == 0 else 0

    def _check(self, rc):
        if rc != 0:
            from .hw_client import raise_for
            raise_for(rc, (self._L.havac_gather_last_error(self._h) or b"").decode())

    def counts(self, my_count: int, stream: int = 0):
        out = (C.c_int64 * self.world)()
        self._check(self._L.havac_gather_counts(self._h, my_count, out, stream or None))
        return list(out)

    def records(self, d_records: int, d_out: int = 0, out_capacity: int = 0, stream: int = 0):
        self._check(self._L.havac_gather_records(self._h, d_records or None, d_out or None, out_capacity, stream or None))

    def set_deadline(self, timeout_ms: int):
        self._check(self._L.havac_gather_set_deadline(self._h, int(timeout_ms)))

    def wait(self):
        """for what the last records() enqueued, with the deadline"""
        self._check(self._L.havac_gather_wait(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.havac_gather_destroy(self._h)
            self._h = None


_c_gathers = {}     # process group -> RcclGather (a communicator is made once per process and group)


def c_gather(group=None):
    """The process's RcclGather for `group`, made on first use (collective: every rank of the group calls this at the same
    point): rank 0 draws the id, torch.distributed carries it to the others."""
    key = id(group) if group is not None else None
    if key in _c_gathers:
        return _c_gathers[key]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    # Every step is agreed on by all ranks before the next one (a MIN all-reduce of "it worked here" through torch.distributed):
    # a rank that cannot load librccl or make its communicator must not leave the others inside ncclCommInitRank or, later,
    # inside a gather it never joins.  Where any rank fails, every rank takes torch.distributed's point-to-point route
    # (gather_hits) for good -- the same exchange, the same buffers.
    def agreed(ok: bool) -> bool:
        on_gpu = dist.get_backend(group) == "nccl"
        if on_gpu:      # (torch's communicator and the library's own must never have operations in flight side by side)
            torch.cuda.synchronize()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(flag.item())
    g, uid, why = None, None, ""
    try:
        uid = RcclGather.unique_id() if rank == 0 else None
    except Exception as e:      # noqa: BLE001 -- whatever went wrong, the other ranks must hear of it
        why = str(e)
    box = [uid]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if agreed(box[0] is not None):
        try:
            g = RcclGather(rank, world, box[0])
            if _deadline_ms:
                g.set_deadline(_deadline_ms)
        except Exception as e:      # noqa: BLE001
            why = str(e)
        if not agreed(g is not None):
            if g is not None:
                g.close()
            g = None
    if g is None and rank == 0:
        import sys
        print(f"havac_amd.dist: the C-ABI gather is not available on every rank ({why or 'another rank failed'}); "
              "the records travel through torch.distributed's point-to-point operations instead", file=sys.stderr, flush=True)
    _c_gathers[key] = g
    return g


def close_c_gathers():
    """collective (ncclCommDestroy): before the process group goes away"""
    for g in _c_gathers.values():
        if g is not None:
            g.close()
    _c_gathers.clear()


def _gather_hits_c(local_hits: torch.Tensor, local_count: int, group, out):
    """gather_hits over havac_gather_* (the nccl backend's path): counts by one all-gather, then grouped send / receive of
    exactly count[r] records into ONE buffer on rank 0, all of it enqueued by libhavac_dev.so behind the current stream."""
    g = c_gather(group)
    stream = torch.cuda.current_stream(local_hits.device).cuda_stream
    counts = g.counts(local_count, stream)
    bad = [r for r, c in enumerate(counts) if c < 0]
    if bad:
        raise ShardFailure(f"the pass failed on rank(s) {bad}; no records were gathered")
    total = sum(counts)
    if g.rank != 0:
        g.records(local_hits.data_ptr(), 0, 0, stream)
        return None, counts
    if out is not None and out.device == local_hits.device and out.numel() >= total:
        merged = out[:total]
    else:
        merged = torch.empty(total, dtype=local_hits.dtype, device=local_hits.device)
    g.records(local_hits.data_ptr(), merged.data_ptr(), merged.numel(), stream)
    return merged, counts


class ShardFailure(RuntimeError):
    """Some rank's pass failed; raised on EVERY rank after the count exchange."""


def gather_hits(local_hits: torch.Tensor, local_count: int, group=None, out: torch.Tensor | None = None):
    """All ranks call this with their own records (int64 view of the packed u64, first
    `local_count` valid; `local_count` = FAILED if this rank's pass raised).  Returns
    (records concatenated in rank order, per-rank counts) on rank 0 and (None, counts)
    elsewhere.  `out`: optional receive buffer on rank 0 (used when it is large enough)."""
    if _c_route(group) and c_gather(group) is not None:
        # the records travel through libhavac_dev.so's own RCCL calls (set_gather_route("torch"): the same exchange through
        # torch.distributed's point-to-point operations, kept for comparison and taken when the C route cannot be set up
        # on every rank)
        return _gather_hits_c(local_hits, local_count, group, out)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out_dev = local_hits.device
    # gloo (CPU rehearsals and the tests) moves host tensors; nccl = RCCL moves device tensors over xGMI
    dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else out_dev
    if dev != out_dev:
        local_hits = local_hits[: max(local_count, 0)].to(dev)
    counts_t = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts_t, torch.tensor([local_count], dtype=torch.int64, device=dev), group=group)
    counts = counts_t.tolist()       # the one host wait of the gather; with passes in flight it waits on the slot's
    #                                  stream only, while the next pass's kernel is already running on another
    bad = [r for r, c in enumerate(counts) if c < 0]
    if bad:
        raise ShardFailure(f"the pass failed on rank(s) {bad}; no records were gathered")
    total = sum(counts)
    root = dist.get_global_rank(group, 0) if group is not None else 0
    if rank != 0:
        if local_count:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_hits[:local_count], root, group=group)]):
                w.wait()
        return None, counts
    if out is not None and out.device == dev and out.numel() >= total:
        merged = out[:total]
    else:
        merged = torch.empty(total, dtype=local_hits.dtype, device=dev)
    ops, offset = [], counts[0]
    for r in range(1, world):
        if counts[r]:
            peer = dist.get_global_rank(group, r) if group is not None else r
            ops.append(dist.P2POp(dist.irecv, merged[offset: offset + counts[r]], peer, group=group))
        offset += counts[r]
    works = dist.batch_isend_irecv(ops) if ops else []
    if counts[0]:
        merged[: counts[0]].copy_(local_hits[: counts[0]])
    for w in works:
        w.wait()
    return (merged.to(out_dev) if dev != out_dev else merged), counts


_NO_STREAM = C.c_void_p(-1)      # include/havac_dev.h: HAVAC_NO_STREAM


class _DeviceRecords:
    """`count` packed records at a device address that libhavac_dev.so owns, for torch.as_tensor (no copy)"""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i8", "data": (ptr, False), "version": 2}


_views = {}      # (device address, device) -> (a tensor over the buffer as far as it has been asked for, its length)


def _records_tensor(ptr: int, count: int, device: torch.device, capacity: int = 0) -> torch.Tensor:
    """the first `count` records at `ptr` as a tensor: a slice of one cached view per buffer (making a tensor from a raw address
    costs ~10 us, a slice 2: this sits on the strictly serial path)"""
    if count == 0 or not ptr:
        return torch.empty(0, dtype=torch.int64, device=device)
    key = (ptr, device.index)
    view = _views.get(key)
    if view is None or view.numel() < count:
        if len(_views) > 64:
            _views.clear()
        view = torch.as_tensor(_DeviceRecords(ptr, max(count, capacity)), device=device)     # (capacity: what the buffer is known to hold)
        _views[key] = view
    return view[:count]


class ShardedSsv:
    """Rank-local driver: enqueue this rank's shard, order its hits, gather to rank 0 (already in order).

    A thin binding of libhavac_dev.so's pipe (include/havac_dev.h level 2b, havac_amd/csrc/havac_pipe.hip): the slots -- a
    context, a hit buffer and an ordering stream each --, the kernel streams and the bookkeeping of passes in flight live in
    C++ since round 5, so that a C++ caller of the drop-in API gets the same engine (havac_dev_set_pipeline_depth).
    depth > 1 keeps that many passes in flight: ``submit`` enqueues a whole pass, ``collect`` finishes the oldest one.  While
    the host waits for pass k and its records are ordered and (N > 1) gathered over RCCL, the SSV kernel of pass k+1 is
    already running: a pass -- preparation, SSV kernel, ordering -- is the business of ONE of two high-priority streams, and
    consecutive passes alternate between them (kernel_streams: None = that; 1 = every pass on one stream, experiments).  A
    kernel's event-timed duration then includes what its neighbours took of the chip meanwhile; a depth-1 engine measures a
    kernel alone.

    The records ``collect`` returns live in libhavac_dev.so's buffers (the slot's hit buffer, or rank 0's receive buffer):
    they are valid until that slot is submitted again, and the caller's current stream has been made to wait for them.

    The gather goes through libhavac_dev.so's own RCCL calls (route "c_abi": the nccl backend, or any backend once
    use_gather_library() named a stand-in) -- then it, too, is the pipe's business -- or through torch.distributed (gloo in the
    CPU tests and rehearsals; set_gather_route("torch"))."""

    def __init__(self, hit_capacity: int, device: torch.device, depth: int = 1, gather_when_alone: bool = False, tuning=None,
                 kernel_streams: int | None = None):
        """tuning: optional (rows_per_block, tiles_per_item, block_tails, ordering[, parts_log2, split_rounds_x4, short_rows, guide[, variant]])
        for SsvContext.set_tuning / set_split_tuning / set_kernel_variant (experiments)"""
        from . import _lib
        from .ssv import SsvContext
        if kernel_streams not in (None, 1, 2, 3, 4):
            raise ValueError("kernel_streams: None (the library's rule) or 1 .. 4")
        self._L = _lib.load()
        self.device = device
        self.depth = max(1, depth)
        # rehearsals on one GPU: run the collectives even in a one-rank group
        self.gather_when_alone = gather_when_alone and dist.is_initialized()
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._gathers = self.world > 1 or self.gather_when_alone
        h = C.c_void_p()
        torch.cuda.set_device(device)        # (the pipe lives on the device that is current now; submit / collect make it current themselves)
        rc = self._L.havac_pipe_create(self.depth, hit_capacity, -1 if kernel_streams is None else kernel_streams, C.byref(h))
        if rc != 0:
            from .hw_client import raise_for
            raise_for(rc, "could not create the pipe (no gfx950 device, or out of memory)")
        self._h = h
        self._capacity = hit_capacity
        self.slots = list(range(self.depth))                 # (bench.py looks at len(engine.slots))
        self.in_flight = []                                   # slot numbers, oldest first (a mirror of the pipe's own)
        self._next = 0
        self._contexts = [SsvContext.borrowed(self._L.havac_pipe_context(self._h, k)) for k in range(self.depth)]
        for ctx in self._contexts:
            if tuning:
                ctx.set_tuning(*tuning[:4])
                if len(tuning) > 4:
                    ctx.set_split_tuning(*tuning[4:8])
                if len(tuning) > 8:
                    ctx.set_kernel_variant(tuning[8])
        self.ctx = self._contexts[0]      # the context of the most recently collected pass (for last_ms)
        self._c_route = False
        self._py_streams, self._merged = [None] * self.depth, [None] * self.depth
        if self._gathers:
            g = c_gather(None) if _c_route(None) else None
            if g is not None:
                self._c_route = True
                self._check(self._L.havac_pipe_set_gather(self._h, g._h))
            elif self.depth > 1:      # the torch route: the gather of a slot runs on a stream of its own, at the kernel streams'
                #                       priority (the host waits for its count exchange: havac_gather.hip says why it is not low)
                _low, high = torch.cuda.Stream.priority_range()
                self._py_streams = [torch.cuda.Stream(device, priority=high) for _ in range(self.depth)]
        self.gather_ms = []               # device time of each gather on this rank (filled by gather_times())
        self._timed = []

    def _check(self, rc):
        if rc < 0:
            from .hw_client import raise_for
            raise_for(rc, (self._L.havac_pipe_last_error(self._h) or b"").decode())

    @property
    def used_two_streams(self) -> bool:
        return bool(self._L.havac_pipe_used_two_streams(self._h))

    @property
    def streams_used(self) -> int:
        """how many streams the last submit took its turn over (1: every pass on one stream)"""
        return int(self._L.havac_pipe_streams_used(self._h))

    def set_sequence_window(self, first_column: int = 0, ncolumns: int = 0):
        """the d_seq handed to submit() holds columns [first_column, first_column + ncolumns) of the database only
        (havac_amd.ssv.shard_window tells a rank what it needs: its shard, the left halo, a little for the tiling)"""
        for ctx in self._contexts:
            ctx.set_sequence_window(first_column, ncolumns)

    def submit(self, d_seq: torch.Tensor, nsymbols: int, d_phmm: torch.Tensor, nrows: int, inputs_ready: bool = False):
        """inputs_ready: d_seq and d_phmm are in place (the caller has synchronised since they were written): the pass does not
        wait for the caller's current stream -- an event on torch's default stream costs a short pass several per cent"""
        if len(self.in_flight) == self.depth:
            raise RuntimeError("every slot is in flight: collect() first")
        slot = self._next
        wait_for = _NO_STREAM
        if self._py_streams[slot] is not None or not inputs_ready:
            current = torch.cuda.current_stream(self.device)
            if self._py_streams[slot] is not None:
                current.wait_stream(self._py_streams[slot])          # the slot's last gather (torch route) has read its hit buffer
            wait_for = current.cuda_stream or None
        self._check(self._L.havac_pipe_submit(self._h, d_seq.data_ptr(), nsymbols, d_phmm.data_ptr(), nrows, self.rank, self.world,
                                              None, wait_for))
        self.in_flight.append(slot)
        self._next = (slot + 1) % self.depth

    def collect(self):
        """-> (records on rank 0 in device order or None, hits found by this rank) of the oldest pass in flight.
        If the pass failed on any rank, every rank raises (this rank's own error, or ShardFailure)."""
        slot = self.in_flight.pop(0)
        current = torch.cuda.current_stream(self.device)
        found, ptr, n = C.c_uint64(0), C.c_void_p(), C.c_uint64(0)
        error = None
        rc = self._L.havac_pipe_collect(self._h, C.byref(found), C.byref(ptr), C.byref(n), current.cuda_stream or None)
        self.ctx = self._contexts[slot]
        if rc < 0:
            try:
                self._check(rc)
            except Exception as e:      # noqa: BLE001 -- the torch route still has collectives to take part in
                error = e
        if not self._gathers or self._c_route:
            if error is not None:
                if "the pass failed on rank(s)" in str(error):
                    raise ShardFailure(str(error))
                raise error
            if not self._gathers:
                return _records_tensor(ptr.value, found.value, self.device, self._capacity), found.value
            return (_records_tensor(ptr.value, n.value, self.device) if self.rank == 0 else None), found.value
        # the torch route (gloo rehearsals; set_gather_route("torch")): the pipe ordered this rank's records, torch.distributed moves them
        local = _records_tensor(ptr.value, found.value, self.device, self._capacity) if error is None else torch.empty(0, dtype=torch.int64, device=self.device)
        stream = self._py_streams[slot] if self._py_streams[slot] is not None else current
        try:
            with torch.cuda.stream(stream):
                if stream is not current:
                    stream.wait_stream(current)
                before, after = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                before.record(stream)
                merged, counts = gather_hits(local, FAILED if error is not None else found.value, out=self._merged[slot])
                after.record(stream)
        except ShardFailure:
            if error is not None:
                raise error
            raise
        self._timed.append((before, after))
        if merged is not None:
            if self._merged[slot] is None or merged.numel() > self._merged[slot].numel():
                self._merged[slot] = merged
            if stream is not current:      # hand the records over to the caller's stream
                current.wait_stream(stream)
                merged.record_stream(current)
        return merged, found.value

    def run_many(self, nsteps: int, d_seq: torch.Tensor, nsymbols: int, d_phmm: torch.Tensor, nrows: int, inputs_ready: bool = False):
        """`nsteps` passes of the same inputs, as many in flight as the engine is deep, all complete on return
        -> ((records on rank 0 or None, hits found by this rank) of the last pass, [(kernel ms, enqueue-to-ordered ms)] per pass).
        Where the pipe does the gather itself (or there is none) the loop runs inside libhavac_dev.so (havac_pipe_run): a timed
        region without Python in it, i.e. what a C++ caller of the library gets."""
        if self.in_flight:
            raise RuntimeError("passes are in flight: collect() them first")
        if self._gathers and not self._c_route:      # the torch route gathers in Python, pass by pass
            result, timings = None, []
            for _ in range(nsteps):
                self.submit(d_seq, nsymbols, d_phmm, nrows, inputs_ready)
                if len(self.in_flight) == self.depth:
                    result = self.collect()
                    timings.append(self.ctx.last_ms())
            while self.in_flight:
                result = self.collect()
                timings.append(self.ctx.last_ms())
            return result, timings
        current = torch.cuda.current_stream(self.device)
        k_ms, t_ms = (C.c_float * max(1, nsteps))(), (C.c_float * max(1, nsteps))()
        found, ptr, n = C.c_uint64(0), C.c_void_p(), C.c_uint64(0)
        rc = self._L.havac_pipe_run(self._h, nsteps, d_seq.data_ptr(), nsymbols, d_phmm.data_ptr(), nrows, self.rank, self.world,
                                    _NO_STREAM if inputs_ready and not self._gathers else (current.cuda_stream or None), k_ms, t_ms,
                                    C.byref(found), C.byref(ptr), C.byref(n))
        self._next = (self._next + nsteps) % self.depth
        self.ctx = self._contexts[(self._next - 1) % self.depth]
        if rc < 0:
            try:
                self._check(rc)
            except Exception as e:      # noqa: BLE001
                if "the pass failed on rank(s)" in str(e):
                    raise ShardFailure(str(e))
                raise
        timings = [(k_ms[i], t_ms[i]) for i in range(nsteps)]
        if not self._gathers:
            return (_records_tensor(ptr.value, found.value, self.device, self._capacity), found.value), timings
        return ((_records_tensor(ptr.value, n.value, self.device) if self.rank == 0 else None), found.value), timings

    def wait_gathers(self):
        """The records of the last collected pass may still be travelling (the gather is enqueued, not waited for): waits for
        them with the gather's deadline (set_gather_deadline; hw_client.CollectiveTimeout names rank and stage) -- before a
        device-wide synchronise, which a peer that died would never let return."""
        if self._c_route:
            self._check(self._L.havac_pipe_wait_gathers(self._h))

    def gather_times(self):
        """device milliseconds of every gather so far (synchronises the events)"""
        if self._c_route:
            n = C.c_uint32(0)
            buf = (C.c_float * 4096)()
            self._check(self._L.havac_pipe_gather_times(self._h, buf, 4096, C.byref(n)))
            self.gather_ms.extend(buf[: min(n.value, 4096)])
        for before, after in self._timed:
            after.synchronize()
            self.gather_ms.append(before.elapsed_time(after))
        self._timed = []
        return self.gather_ms

    def run(self, d_seq: torch.Tensor, nsymbols: int, d_phmm: torch.Tensor, nrows: int):
        """one pass, start to end -> (records on rank 0 in device order or None, hits found by this rank)"""
        self.submit(d_seq, nsymbols, d_phmm, nrows)
        return self.collect()

    def release(self):
        """Gives the device memory of every slot back: contexts with their ordering buffers, hit buffers, receive buffers.
        The records collect() returned last stay valid until close(); everything else is freed now.  Nothing may be in
        flight, and nothing can be submitted afterwards."""
        if self.in_flight:
            raise RuntimeError("passes are in flight: collect() them first")
        self._check(self._L.havac_pipe_release(self._h))
        self._merged = [None] * self.depth
        _views.clear()
        torch.cuda.empty_cache()

    def close(self):
        if getattr(self, "_h", None):
            self._L.havac_pipe_destroy(self._h)
            self._h = None

    __del__ = close
