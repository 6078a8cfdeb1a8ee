"""ctypes front end of the CPU checker.  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product package ``havac_amd`` never
does (tests/test_boundary.py greps for it).

Two libraries are wrapped:

* ``oracle/liboracle.so``      -- our restatement (``ssv_oracle.c``)
* ``oracle/_ref/libsoftssv_ref.so`` -- the reference's own ``softSsvThreshold256``
  (built in the build container only, see ``oracle/Makefile``); optional.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SEGMENT = 12288

_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


def build(with_ref: bool = True) -> None:
    """Compile liboracle.so and, when /root/reference is present, _ref/."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)
    if with_ref and os.path.isfile("/root/reference/test/softSsv/SoftSsv.cpp"):
        subprocess.run(["make", "-C", _HERE, "_ref"], check=True, capture_output=True)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.isfile(path):
            build(with_ref=False)
        L = C.CDLL(path)
        L.havac_oracle_cell.restype = C.c_uint8
        L.havac_oracle_cell.argtypes = [C.c_uint8, C.c_int8, C.POINTER(C.c_int)]
        L.havac_oracle_cell_carry.restype = C.c_uint8
        L.havac_oracle_cell_carry.argtypes = [C.c_uint8, C.c_uint8, C.POINTER(C.c_int)]
        L.havac_oracle_pack_hit.restype = C.c_uint64
        L.havac_oracle_pack_hit.argtypes = [C.c_uint32, C.c_uint64]
        L.havac_oracle_unpack_2bit.argtypes = [_u8p, C.c_uint64, _u8p]
        L.havac_oracle_pack_2bit.argtypes = [_u8p, C.c_uint64, _u8p]
        L.havac_oracle_ssv.restype = C.c_int64
        L.havac_oracle_ssv.argtypes = [_u8p, C.c_uint64, _i8p, C.c_uint64, _u64p, C.c_uint64]
        L.havac_oracle_ssv_window.restype = C.c_int64
        L.havac_oracle_ssv_window.argtypes = [_u8p, C.c_uint64, _i8p, C.c_uint64, C.c_uint64,
                                              C.c_uint64, _u64p, C.c_uint64]
        L.havac_oracle_ssv_mt.restype = C.c_int64
        L.havac_oracle_ssv_mt.argtypes = [_u8p, C.c_uint64, _i8p, C.c_uint64, _u64p, C.c_uint64, C.c_int]
        L.havac_oracle_ssv_fast.restype = C.c_int64
        L.havac_oracle_ssv_fast.argtypes = [_u8p, C.c_uint64, _i8p, C.c_uint64, _u64p, C.c_uint64, C.c_int]
        L.havac_oracle_cells.restype = C.c_int
        L.havac_oracle_cells.argtypes = [_u8p, C.c_uint64, _i8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _u8p]
        L.havac_oracle_sort_device_order.argtypes = [_u64p, C.c_uint64]
        L.havac_oracle_sort_row_major.argtypes = [_u64p, C.c_uint64]
        _lib = L
    return _lib


def build_info(reference: bool = False) -> str:
    """'<compiler> <version> <flags>' of liboracle.so, or of _ref/libsoftssv_ref.so (the reference's softSsv)"""
    L = ref() if reference else lib()
    fn = getattr(L, "softssv_ref_build_info" if reference else "havac_oracle_build_info", None)
    if fn is None:          # a library built before the symbol existed (a stale .so is not rebuilt by lib() / ref())
        return "unknown (a stale checker library: run `make -C oracle all _ref`)"
    fn.restype = C.c_char_p
    fn.argtypes = []
    return fn().decode()


def ref_available() -> bool:
    return os.path.isfile(os.path.join(_HERE, "_ref", "libsoftssv_ref.so"))


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(os.path.join(_HERE, "_ref", "libsoftssv_ref.so"))
        R.softssv_ref_run.restype = C.c_int64
        R.softssv_ref_run.argtypes = [_u8p, C.c_uint64, _i8p, C.c_uint64, _u64p, C.c_uint64]
        _ref = R
    return _ref


# ---------------------------------------------------------------------------

def unpack_2bit(packed: np.ndarray, nsymbols: int | None = None) -> np.ndarray:
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n = packed.size * 4 if nsymbols is None else nsymbols
    out = np.empty(n, dtype=np.uint8)
    lib().havac_oracle_unpack_2bit(packed, n, out)
    return out


def pack_2bit(symbols: np.ndarray) -> np.ndarray:
    symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
    out = np.empty((symbols.size + 3) // 4, dtype=np.uint8)
    lib().havac_oracle_pack_2bit(symbols, symbols.size, out)
    return out


def pack_hits(rows, cols) -> np.ndarray:
    rows = np.asarray(rows, dtype=np.uint64)
    cols = np.asarray(cols, dtype=np.uint64)
    seg = cols // np.uint64(SEGMENT)
    inseg = cols % np.uint64(SEGMENT)
    return inseg | (seg << np.uint64(14)) | (rows << np.uint64(40))


def unpack_hits(records: np.ndarray):
    """-> (rows, columns) as uint64 arrays.  host/Havac.cpp:155-163."""
    records = np.asarray(records, dtype=np.uint64)
    inseg = records & np.uint64(0x3FFF)
    seg = (records >> np.uint64(14)) & np.uint64(0x3FFFFFF)
    return records >> np.uint64(40), seg * np.uint64(SEGMENT) + inseg


def device_order(records: np.ndarray) -> np.ndarray:
    out = np.ascontiguousarray(records, dtype=np.uint64).copy()
    lib().havac_oracle_sort_device_order(out, out.size)
    return out


def row_major(records: np.ndarray) -> np.ndarray:
    out = np.ascontiguousarray(records, dtype=np.uint64).copy()
    lib().havac_oracle_sort_row_major(out, out.size)
    return out


def _run(fn, symbols, model, extra=(), cap=None):
    symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
    model = np.ascontiguousarray(model, dtype=np.int8).reshape(-1)
    nrows = model.size // 4
    cap = int(cap if cap is not None else max(1024, symbols.size // 64))
    while True:
        buf = np.empty(cap, dtype=np.uint64)
        n = fn(symbols, symbols.size, model, nrows, *extra, buf, cap)
        if n < 0:
            raise MemoryError("oracle could not allocate its row buffer")
        if n <= cap:
            return buf[:n].copy()
        cap = int(n)


def ssv(symbols, model, cap=None) -> np.ndarray:
    """Whole-matrix oracle; packed records in DEVICE order."""
    return device_order(_run(lib().havac_oracle_ssv, symbols, model, cap=cap))


def ssv_window(symbols, model, col_begin, col_end, cap=None) -> np.ndarray:
    hits = _run(lib().havac_oracle_ssv_window, symbols, model, (int(col_begin), int(col_end)), cap=cap)
    return device_order(hits)


CELL_RECORD = np.dtype([("prev", np.uint8), ("match", np.int8), ("score", np.uint8), ("hit", np.uint8), ("symbol", np.uint8),
                        ("pending", np.uint8), ("zero", np.uint8), ("written", np.uint8)])      # include/havac_dev.h, havac_cell_record


def cells(symbols, model, row0, col0, h, w) -> np.ndarray:
    """per-cell records of rows [row0, row0+h) x columns [col0, col0+w) -> array [h, w] of CELL_RECORD"""
    symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
    model = np.ascontiguousarray(model, dtype=np.int8).reshape(-1)
    out = np.zeros(h * w * 8, dtype=np.uint8)
    rc = lib().havac_oracle_cells(symbols, symbols.size, model, model.size // 4, row0, col0, h, w, out)
    if rc != 0:
        raise ValueError(f"havac_oracle_cells: {rc}")
    return out.view(CELL_RECORD).reshape(h, w)


def ssv_mt(symbols, model, nthreads=0, cap=None) -> np.ndarray:
    symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
    model = np.ascontiguousarray(model, dtype=np.int8).reshape(-1)
    cap = int(cap if cap is not None else max(1024, symbols.size // 64))
    while True:
        buf = np.empty(cap, dtype=np.uint64)
        n = lib().havac_oracle_ssv_mt(symbols, symbols.size, model, model.size // 4, buf, cap, nthreads)
        if n < 0:
            raise MemoryError("oracle could not allocate")
        if n <= cap:
            return buf[:n].copy()
        cap = int(n)


def ssv_fast(symbols, model, nthreads=0, cap=None) -> np.ndarray:
    """AVX2 + threads; the whole matrix, records in DEVICE order.  For full-size comparisons."""
    symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
    model = np.ascontiguousarray(model, dtype=np.int8).reshape(-1)
    cap = int(cap if cap is not None else max(1024, symbols.size // 64))
    while True:
        buf = np.empty(cap, dtype=np.uint64)
        n = lib().havac_oracle_ssv_fast(symbols, symbols.size, model, model.size // 4, buf, cap, nthreads)
        if n < 0:
            raise MemoryError("oracle could not allocate")
        if n <= cap:
            return buf[:n].copy()
        cap = int(n)


def ssv_reference(symbols, model, cap=None) -> np.ndarray:
    """The REFERENCE's softSsvThreshold256; packed records in DEVICE order."""
    symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
    model = np.ascontiguousarray(model, dtype=np.int8).reshape(-1)
    cap = int(cap if cap is not None else max(1024, symbols.size // 64))
    while True:
        buf = np.empty(cap, dtype=np.uint64)
        n = ref().softssv_ref_run(symbols, symbols.size, model, model.size // 4, buf, cap)
        if n < 0:
            raise MemoryError("reference softSsv reported errorCode -1")
        if n <= cap:
            rc = buf[:n]
            return device_order(pack_hits(rc >> np.uint64(32), rc & np.uint64(0xFFFFFFFF)))
        cap = int(n)


def ssv_reference_raw(symbols, model, cap=1 << 20):
    """(rows, cols) in the reference's own emission order."""
    symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
    model = np.ascontiguousarray(model, dtype=np.int8).reshape(-1)
    buf = np.empty(cap, dtype=np.uint64)
    n = ref().softssv_ref_run(symbols, symbols.size, model, model.size // 4, buf, cap)
    assert 0 <= n <= cap
    rc = buf[:n]
    return (rc >> np.uint64(32)).astype(np.uint32), (rc & np.uint64(0xFFFFFFFF)).astype(np.uint32)
