"""Kernel-ABI wrapper (level 2 of include/havac_dev.h): SSV on caller-owned device memory.

Takes raw device addresses (ints) and a raw HIP stream handle so that it stays
independent of torch; bench.py and havac_amd/dist.py pass `tensor.data_ptr()`
and `torch.cuda.current_stream().cuda_stream`.
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from .hw_client import raise_for


class SsvContext:
    def __init__(self):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.havac_ssv_ctx_create(C.byref(h))
        if rc != 0:
            raise_for(rc, "could not create an SSV context (no gfx950 device?)")
        self._h = h

    @classmethod
    def borrowed(cls, handle):
        """a context that somebody else owns (a slot of a pipe: havac_pipe_context); close() leaves it alone"""
        self = cls.__new__(cls)
        self._L = _lib.load()
        self._h = C.c_void_p(handle)
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._L.havac_ssv_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise_for(rc, (self._L.havac_ssv_ctx_last_error(self._h) or b"").decode())

    def enqueue(self, d_sequence: int, nsymbols: int, d_phmm: int, nrows: int, d_hits: int, hit_capacity: int,
                shard_index: int = 0, shard_count: int = 1, d_abort_flag: int = 0, stream: int = 0):
        self._check(self._L.havac_ssv_enqueue(self._h, d_sequence, nsymbols, d_phmm, nrows, shard_index, shard_count,
                                              d_hits, hit_capacity, d_abort_flag or None, stream or None))

    def set_sequence_window(self, first_column: int = 0, ncolumns: int = 0):
        """the d_sequence handed to the next passes holds columns [first_column, first_column + ncolumns) only (0, 0: all)"""
        self._check(self._L.havac_ssv_set_sequence_window(self._h, first_column, ncolumns))

    def set_order_stream(self, stream: int = 0):
        """the HIP stream finish() orders the records on (0 = the stream of the enqueue)"""
        self._check(self._L.havac_ssv_set_order_stream(self._h, stream or None))

    def set_cell_trace(self, d_cells: int = 0, row0: int = 0, col0: int = 0, nrows: int = 0, ncols: int = 0):
        """per-cell trace window of the next passes (include/havac_dev.h, havac_cell_record: nrows * ncols records of
        8 bytes at device address d_cells, cleared by the caller); d_cells = 0 switches it off"""
        self._check(self._L.havac_ssv_set_cell_trace(self._h, d_cells or None, row0, col0, nrows, ncols))

    def finish(self) -> int:
        n = C.c_uint64(0)
        self._check(self._L.havac_ssv_finish(self._h, C.byref(n)))
        return n.value

    def finish_begin(self):
        """first half of finish(): waits for the hit count, enqueues the ordering (begin all contexts, then end all)"""
        self._check(self._L.havac_ssv_finish_begin(self._h))

    def finish_end(self) -> int:
        n = C.c_uint64(0)
        self._check(self._L.havac_ssv_finish_end(self._h, C.byref(n)))
        return n.value

    def set_tuning(self, rows_per_block: int = -1, tiles_per_item: int = -1, block_tails: int = -1, ordering: int = -1):
        """experiment knobs of the next passes (include/havac_dev.h: havac_ssv_set_tuning); -1 = the library's own rule"""
        self._check(self._L.havac_ssv_set_tuning(self._h, rows_per_block, tiles_per_item, block_tails, ordering))

    def set_split_tuning(self, parts_log2: int = -1, split_rounds_x4: int = -1, short_rows: int = -1, guide: int = -1):
        """the library's rule for tall tiles (include/havac_dev.h: havac_ssv_set_split_tuning); -1 = the default"""
        self._check(self._L.havac_ssv_set_split_tuning(self._h, parts_log2, split_rounds_x4, short_rows, guide))

    def set_kernel_variant(self, variant: int = -1):
        """which instantiation of the SSV kernel the next passes run (include/havac_dev.h: havac_ssv_set_kernel_variant):
        -1 the library decides, 0 the standard kernel, 1 the resident-table kernel (short models) wherever it is valid"""
        self._check(self._L.havac_ssv_set_kernel_variant(self._h, variant))

    def last_kernel_variant(self) -> int:
        v = C.c_int(0)
        self._check(self._L.havac_ssv_last_kernel_variant(self._h, C.byref(v)))
        return v.value

    def wave_slots(self) -> int:
        """waves the context's device holds at once: what enqueue() hands to the planner (launch_plan(wave_slots=...))"""
        n = C.c_uint32(0)
        self._check(self._L.havac_ssv_wave_slots(self._h, C.byref(n)))
        return n.value

    def last_ordering(self):
        """-> (path, buckets, largest bucket) of the last finished pass: path 0 radix sort, 1 bucket ordering, 2 fallback"""
        path, nb, big = C.c_int(0), C.c_uint32(0), C.c_uint32(0)
        self._check(self._L.havac_ssv_last_ordering(self._h, C.byref(path), C.byref(nb), C.byref(big)))
        return path.value, nb.value, big.value

    def sort_hits(self, d_hits: int, count: int, stream: int = 0):
        self._check(self._L.havac_ssv_sort_hits(self._h, d_hits, count, stream or None))

    def last_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        self._check(self._L.havac_ssv_last_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value


class _OrderReport(C.Structure):
    _fields_ = [("records", C.c_uint64), ("out_of_order", C.c_uint64), ("first_out_of_order", C.c_uint64),
                ("out_of_span", C.c_uint64), ("first_out_of_span", C.c_uint64)]


def check_order(d_records: int, count: int, rank_counts=(), spans=(), stream: int = 0) -> dict:
    """The check of a gathered list on the device (include/havac_dev.h: havac_ssv_check_order): `count` packed records at
    device address `d_records`, the ranks' lists end to end; rank r contributed rank_counts[r] records and owns columns
    spans[r] = (begin, end).  One pass, no list-sized temporary -> dict(records, out_of_order, first_out_of_order,
    out_of_span, first_out_of_span); the two `first_*` are None when there is no such record."""
    n = len(rank_counts)
    if len(spans) != n:
        raise ValueError("one column span per rank")
    u64 = C.c_uint64 * max(1, n)
    counts = u64(*[int(c) for c in rank_counts])
    begin = u64(*[int(s[0]) for s in spans])
    end = u64(*[int(s[1]) for s in spans])
    rep = _OrderReport()
    rc = _lib.load().havac_ssv_check_order(d_records or None, count, counts, begin, end, n, stream or None, C.byref(rep))
    if rc != 0:
        raise_for(rc, "the ranks' counts do not add up to the list" if rc == _lib.E_LENGTH else "havac_ssv_check_order failed")
    none = (1 << 64) - 1
    return {"records": rep.records, "out_of_order": rep.out_of_order,
            "first_out_of_order": None if rep.first_out_of_order == none else rep.first_out_of_order,
            "out_of_span": rep.out_of_span, "first_out_of_span": None if rep.first_out_of_span == none else rep.first_out_of_span}


def shard_cells(nsymbols: int, nrows: int, shard_index: int = 0, shard_count: int = 1) -> int:
    return int(_lib.load().havac_ssv_shard_cells(nsymbols, nrows, shard_index, shard_count))


def shard_columns(nsymbols: int, shard_index: int, shard_count: int):
    """-> (col_begin, col_end): the whole 12288-column segments shard `shard_index` of `shard_count` reports."""
    b, e = C.c_uint64(0), C.c_uint64(0)
    rc = _lib.load().havac_ssv_shard_columns(nsymbols, shard_index, shard_count, C.byref(b), C.byref(e))
    if rc != 0:
        raise_for(rc, "bad shard arguments")
    return b.value, e.value


def shard_window(nsymbols: int, nrows: int, shard_index: int, shard_count: int):
    """-> (first_column, end_column): the whole segments shard `shard_index` of `shard_count` READS (its own columns, the
    left halo of nrows - 1, a few thousand more for the tiling); what a rank must hold of the database."""
    b, e = C.c_uint64(0), C.c_uint64(0)
    rc = _lib.load().havac_ssv_shard_window(nsymbols, nrows, shard_index, shard_count, C.byref(b), C.byref(e))
    if rc != 0:
        raise_for(rc, "bad shard arguments")
    return b.value, e.value


class _LaunchPlan(C.Structure):
    _fields_ = [("nrows_padded", C.c_uint32), ("tile_begin", C.c_uint32), ("ntiles", C.c_uint32),
                ("nparts", C.c_uint32), ("part_begin", C.c_uint32 * 9),
                ("tiles_per_group", C.c_uint32), ("single_tiles", C.c_uint32), ("cut_tiles", C.c_uint32),
                ("nrow_blocks", C.c_uint32), ("ncuts", C.c_uint32), ("uniform_rows", C.c_uint32), ("row_cut", C.c_uint32 * 33),
                ("workgroups", C.c_uint32), ("resident_kernel", C.c_uint32), ("walk_slots", C.c_uint32), ("walk_rounds", C.c_uint32),
                ("walk_len", C.c_uint32 * 8), ("walk_base", C.c_uint32 * 8)]


def launch_plan(nsymbols: int, nrows: int, shard_index: int = 0, shard_count: int = 1, wave_slots: int = 0, tuning=()):
    """How a launch of this shape hands out its tiles (include/havac_dev.h: havac_ssv_plan; no device needed) -> dict with
    the partitions, the grouping of short models' tiles, the row blocks of cut tiles, and `items`: per partition the list of
    (first tile of the launch, tiles walked, row block or None) in the order its workgroups take them -- for a launch of the
    resident-table kernel (short models) one "partition" with one item per wave: its run of adjacent tiles."""
    plan = _LaunchPlan()
    arr = (C.c_int32 * max(1, len(tuning)))(*[int(v) for v in tuning])
    rc = _lib.load().havac_ssv_plan(nsymbols, nrows, shard_index, shard_count, wave_slots, arr, len(tuning), C.byref(plan))
    if rc != 0:
        raise_for(rc, "no plan for this shape / tuning")
    out = {name: getattr(plan, name) for name, _ in _LaunchPlan._fields_ if name not in ("part_begin", "row_cut", "walk_len", "walk_base")}
    out["walk_len"] = list(plan.walk_len)[: plan.walk_rounds] if plan.resident_kernel else []
    out["walk_base"] = list(plan.walk_base)[: plan.walk_rounds] if plan.resident_kernel else []
    out["part_begin"] = list(plan.part_begin)[: plan.nparts + 1]
    cuts = list(plan.row_cut)[: plan.ncuts + 1]
    blocks = [(cuts[b], cuts[b + 1]) for b in range(plan.ncuts)]
    at = cuts[-1] if cuts else 0
    while len(blocks) < plan.nrow_blocks:
        blocks.append((at, min(at + plan.uniform_rows, plan.nrows_padded)))
        at += plan.uniform_rows
    out["row_blocks"] = blocks if plan.nrow_blocks > 1 else []
    items = []
    if plan.resident_kernel:
        runs = []
        for g in range(plan.workgroups * 4):
            r = min(g // plan.walk_slots, plan.walk_rounds - 1)
            first = plan.walk_base[r] + (g - r * plan.walk_slots) * plan.walk_len[r]
            first, end = min(first, plan.ntiles), min(first + plan.walk_len[r], plan.ntiles)
            if end > first:
                runs.append((first, end - first, None))
        out["items"] = [runs]
        return out
    for k in range(plan.nparts):
        t0, t1 = plan.part_begin[k], plan.part_begin[k + 1]
        mine = t1 - t0
        cut = min(plan.cut_tiles, mine) if plan.nrow_blocks > 1 else 0
        whole = mine - cut
        g = plan.tiles_per_group
        groups = (whole - min(plan.single_tiles, whole)) // g if g > 1 else 0
        part = [(t0 + i * g, g, None) for i in range(groups)]
        part += [(t, 1, None) for t in range(t0 + groups * g, t0 + whole)]
        part += [(t0 + whole + j, 1, b) for b in range(plan.nrow_blocks if cut else 0) for j in range(cut)]
        items.append(part)
    out["items"] = items
    return out
