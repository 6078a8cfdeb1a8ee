"""Python mirror of the reference's `HavacHwClient` on top of the C ABI.

Same method names, argument meaning and error behaviour as
host/HavacHwClient.hpp:22-75 / host/HavacHwClient.cpp:25-202, so that the
parity tests read like the reference's own.  All work happens in
libhavac_dev.so on the GPU; this file only moves numpy buffers across ctypes
and maps error codes back to the exception types the reference throws.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class LengthError(ValueError):
    """std::length_error (host/HavacHwClient.cpp:88,96,118,124,143,146)."""


class LogicError(RuntimeError):
    """std::logic_error (host/HavacHwClient.cpp:168)."""


class HitOverflowError(RuntimeError):
    """HAVAC_E_HIT_OVERFLOW: more hits than the hit buffer holds."""


class NoDeviceError(RuntimeError):
    """No usable gfx950 device."""


class CollectiveTimeout(RuntimeError):
    """HAVAC_E_TIMEOUT: a collective of a sharded run did not complete before its deadline (havac_gather_set_deadline);
    the message names the rank and the stage."""


def raise_for(code: int, message: str):
    if code >= 0:
        return
    if code == _lib.E_LENGTH:
        raise LengthError(message)
    if code == _lib.E_LOGIC:
        raise LogicError(message)
    if code == _lib.E_NOMEM:
        raise MemoryError(message)
    if code == _lib.E_HIT_OVERFLOW:
        raise HitOverflowError(message)
    if code == _lib.E_NO_DEVICE:
        raise NoDeviceError(message or "no gfx950 device")
    if code == _lib.E_ARGUMENT:
        raise ValueError(message or "bad argument")
    if code == _lib.E_TIMEOUT:
        raise CollectiveTimeout(message or "a collective did not complete before its deadline")
    raise RuntimeError(message or f"havac_dev error {code}")


class HavacHwClient:
    """HavacHwClient(xclbinFileSrc, havacKernelName, deviceIndex) -- the first two are accepted and
    ignored: there is no bitstream, the kernels live in libhavac_dev.so."""

    def __init__(self, xclbinFileSrc: str = "", havacKernelName: str = "HavacKernel", deviceIndex: int = 0,
                 deviceIndices=None):
        """deviceIndices (an addition): several GPUs behind one client, one column shard each."""
        self._L = _lib.load()
        h = C.c_void_p()
        if deviceIndices is not None:
            arr = (C.c_uint32 * len(deviceIndices))(*deviceIndices)
            rc = self._L.havac_dev_create_multi(arr, len(deviceIndices), C.byref(h))
        else:
            rc = self._L.havac_dev_create(deviceIndex, C.byref(h))
        if rc != 0:
            raise_for(rc, "could not create the device client")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.havac_dev_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise_for(rc, (self._L.havac_dev_last_error(self._h) or b"").decode())
        return rc

    def setHitCapacity(self, max_hits: int):
        self._check(self._L.havac_dev_set_hit_capacity(self._h, int(max_hits)))

    def setPipelineDepth(self, depth: int):
        """Not in the reference (include/havac_dev.h: havac_dev_set_pipeline_depth): up to `depth` runs open at once; the other
        methods speak of the oldest open run, retire() closes it."""
        self._check(self._L.havac_dev_set_pipeline_depth(self._h, int(depth)))

    def retire(self):
        self._check(self._L.havac_dev_retire(self._h))

    def openRuns(self) -> int:
        return int(self._L.havac_dev_open_runs(self._h))

    def setTuning(self, *values: int):
        """experiment knobs (include/havac_dev.h: havac_dev_set_tuning): rows_per_block, tiles_per_item, block_tails, ordering,
        parts_log2, split_rounds_x4, short_rows, guide, kernel variant (0 standard, 1 short-model); -1 or missing = the library's own rule"""
        import ctypes
        arr = (ctypes.c_int32 * len(values))(*[int(v) for v in values])
        self._check(self._L.havac_dev_set_tuning(self._h, arr, len(values)))

    def writeSequence(self, compressedSequence):
        buf = np.ascontiguousarray(compressedSequence, dtype=np.uint8)
        self._check(self._L.havac_dev_write_sequence(self._h, buf.ctypes.data, buf.size))

    def writeSequenceChars(self, chars, patchColumns=(), patchSymbols=()):
        """Not in the reference: the sequence as text, one byte per column, packed on the
        GPU; `patchColumns` (ascending) / `patchSymbols` (0..3) give the columns that are not a/c/g/t."""
        buf = np.ascontiguousarray(np.frombuffer(chars, dtype=np.uint8) if isinstance(chars, (bytes, bytearray)) else chars,
                                   dtype=np.uint8)
        cols = np.ascontiguousarray(patchColumns, dtype=np.uint64)
        syms = np.ascontiguousarray(patchSymbols, dtype=np.uint8)
        if cols.size != syms.size:
            raise ValueError("one symbol per patch column")
        self._check(self._L.havac_dev_write_sequence_chars(self._h, buf.ctypes.data if buf.size else None, buf.size,
                                                           cols.ctypes.data if cols.size else None,
                                                           syms.ctypes.data if syms.size else None, cols.size))

    def readSequence(self, nbytes: int) -> np.ndarray:
        """The packed sequence as it stands in HBM (checking aid, not in the reference)."""
        out = np.empty(int(nbytes), np.uint8)
        self._check(self._L.havac_dev_read_sequence(self._h, out.ctypes.data if out.size else None, out.size))
        return out

    def writeSequenceRecords(self, chars, recordEnds):
        """Not in the reference (rows f2 + f4): the boundary-mode layout made on the GPU from the text; returns each
        record's first column."""
        buf = np.ascontiguousarray(np.frombuffer(chars, dtype=np.uint8) if isinstance(chars, (bytes, bytearray)) else chars,
                                   dtype=np.uint8)
        ends = np.ascontiguousarray(recordEnds, dtype=np.uint64)
        starts = np.zeros(ends.size, np.uint64)
        self._check(self._L.havac_dev_write_sequence_records(self._h, buf.ctypes.data if buf.size else None, buf.size,
                                                             ends.ctypes.data if ends.size else None, ends.size,
                                                             starts.ctypes.data if ends.size else None))
        return starts

    def appendReverseStrand(self, starts, residues) -> int:
        """Not in the reference (row f3): doubles the sequence on the device; returns the forward half's columns."""
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        rs = np.ascontiguousarray(residues, dtype=np.uint64)
        nf = C.c_uint64(0)
        self._check(self._L.havac_dev_append_reverse_strand(self._h, st.ctypes.data if st.size else None,
                                                            rs.ctypes.data if rs.size else None, st.size, C.byref(nf)))
        return nf.value

    def readSeparatorMask(self, nbytes: int) -> np.ndarray:
        out = np.empty(int(nbytes), np.uint8)
        self._check(self._L.havac_dev_read_separator_mask(self._h, out.ctypes.data if out.size else None, out.size))
        return out

    def writeSeparatorMask(self, pairBitmap):
        """Boundary mode (not in the reference): one bit per aligned symbol pair; None or empty removes the mask."""
        buf = np.ascontiguousarray(pairBitmap if pairBitmap is not None else [], dtype=np.uint8)
        self._check(self._L.havac_dev_write_separator_mask(self._h, buf.ctypes.data if buf.size else None, buf.size))

    def writePhmm(self, phmmAsFlattenedArray):
        buf = np.ascontiguousarray(phmmAsFlattenedArray, dtype=np.int8).reshape(-1)
        self._check(self._L.havac_dev_write_phmm(self._h, buf.ctypes.data, buf.size))

    def invokeHavacSsvAsync(self):
        self._check(self._L.havac_dev_run_async(self._h))

    def getHwState(self) -> int:
        return self._check(self._L.havac_dev_state(self._h))

    def waitForHavacSsvAsync(self, timeout_ms: int = 0) -> int:
        return self._check(self._L.havac_dev_wait(self._h, int(timeout_ms)))

    def abort(self) -> int:
        return self._check(self._L.havac_dev_abort(self._h))

    def getNumHits(self) -> int:
        n = C.c_uint64(0)          # the 64-bit entry point: several GPUs behind one handle can exceed 2^32 - 1 records
        self._check(self._L.havac_dev_num_hits64(self._h, C.byref(n)))
        return n.value

    def getHitList(self) -> np.ndarray:
        n = self.getNumHits()
        out = np.empty(n, dtype=np.uint64)
        if n:
            self._check(self._L.havac_dev_read_hits64(self._h, out.ctypes.data, n))
        return out

    def lastRunMs(self):
        a, b = C.c_float(0), C.c_float(0)
        self._check(self._L.havac_dev_last_run_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value


def run_ssv(packed_sequence, model, device: int = 0, hit_capacity: int | None = None) -> np.ndarray:
    """Convenience: one whole run through the handle API; hits in device order."""
    c = HavacHwClient(deviceIndex=device)
    try:
        if hit_capacity:
            c.setHitCapacity(hit_capacity)
        c.writeSequence(packed_sequence)
        c.writePhmm(model)
        c.invokeHavacSsvAsync()
        c.waitForHavacSsvAsync()
        return c.getHitList()
    finally:
        c.close()
