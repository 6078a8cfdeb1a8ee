"""ctypes view of the C++ `Havac` class (libhavac.so, include/havac_host.h).

`Havac` here has the reference's method names (host/Havac.hpp:42-107); every
call goes straight into the C++ object, which drives the GPU through
libhavac_dev.so.  The host-only helpers at the bottom run the reference's CPU
stages (FASTA packing, model projection, hit resolution) and need no device.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _lib
from .hw_client import raise_for

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libhavac.so")

_vp = C.c_void_p
HOST_SIGNATURES = {
    "havac_host_create": (C.c_int, [C.c_uint32, C.c_float, C.POINTER(C.c_void_p)]),
    "havac_host_create_deferred": (C.c_int, [C.c_uint32, C.c_float, C.POINTER(C.c_void_p)]),
    "havac_host_create_multi": (C.c_int, [C.POINTER(C.c_uint32), C.c_uint32, C.c_float, C.POINTER(C.c_void_p)]),
    "havac_host_destroy": (None, [_vp]),
    "havac_host_load_sequence": (C.c_int, [_vp, C.c_char_p]),
    "havac_host_load_phmm": (C.c_int, [_vp, C.c_char_p]),
    "havac_host_run": (C.c_int, [_vp]),
    "havac_host_run_async": (C.c_int, [_vp]),
    "havac_host_wait": (C.c_int, [_vp]),
    "havac_host_abort": (C.c_int, [_vp]),
    "havac_host_state": (C.c_int, [_vp]),
    "havac_host_set_hit_capacity": (C.c_int, [_vp, C.c_uint64]),
    "havac_host_set_pipeline_depth": (C.c_int, [_vp, C.c_uint32]),
    "havac_host_next_run": (C.c_int, [_vp]),
    "havac_host_set_boundary_mode": (C.c_int, [_vp, C.c_int]),
    "havac_host_set_both_strands": (C.c_int, [_vp, C.c_int]),
    "havac_host_set_device_packing": (C.c_int, [_vp, C.c_int]),
    "havac_host_text_and_patches": (C.c_int, [C.c_char_p, C.c_int64, _vp, C.c_uint64, C.POINTER(C.c_uint64), _vp, _vp,
                                              C.c_uint64, C.POINTER(C.c_uint64)]),
    "havac_host_get_hit_strands": (C.c_int, [_vp, _vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "havac_host_get_hits": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "havac_host_get_raw_hits": (C.c_int, [_vp, _vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "havac_host_last_run_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "havac_host_last_error": (C.c_char_p, [_vp]),
    "havac_host_pack_fasta": (C.c_int, [C.c_char_p, C.c_int64, _vp, C.c_uint64, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "havac_host_pack_fasta_layout": (C.c_int, [C.c_char_p, C.c_int64, C.c_int, C.c_int, _vp, C.c_uint64, C.POINTER(C.c_uint64),
                                               _vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "havac_host_project_hmm": (C.c_int, [C.c_char_p, C.c_float, _vp, C.c_uint64, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint32), _vp, C.c_uint32]),
    "havac_host_read_hmm_emissions": (C.c_int, [C.c_char_p, _vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "havac_host_scaling_factor": (C.c_float, [C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_float]),
    "havac_host_project_score": (C.c_float, [C.c_float, C.c_float]),
    "havac_host_project_model": (C.c_int, [C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_float, _vp, _vp]),
    "havac_host_gumbel_invsurv": (C.c_double, [C.c_double, C.c_double, C.c_double]),
    "havac_host_resolve_hits": (C.c_int, [C.c_char_p, C.c_char_p, _vp, C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint32,
                                          C.POINTER(C.c_uint32)]),
    "havac_host_merge_windows": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint32, C.c_uint32,
                                           _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "havac_host_get_windows": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32,
                                         C.POINTER(C.c_uint32)]),
}

_host = None


def load_host() -> C.CDLL:
    global _host
    if _host is None:
        _lib.load()       # libhavac.so links libhavac_dev.so; fail on that one first, with its message
        if not os.path.isfile(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} is missing: build it with havac_amd/csrc/host/build.sh")
        lib = C.CDLL(HOST_LIB_PATH)
        for name, (res, args) in HOST_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _host = lib
    return _host


@dataclass(frozen=True)
class HavacHit:      # host/Havac.hpp:31-39
    sequencePosition: int
    sequenceIndex: int
    phmmPosition: int
    phmmIndex: int
    reverseStrand: bool = False      # addition, see Havac.setBothStrands

    def toString(self) -> str:
        return (f"sequence ${self.sequenceIndex}, position {self.sequencePosition}; "
                f"phmm #{self.phmmIndex} position {self.phmmPosition}" + (" (reverse strand)" if self.reverseStrand else ""))


def _hits_from_arrays(sp, si, pp, pi):
    return [HavacHit(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(sp, si, pp, pi)]


class Havac:
    def __init__(self, deviceIndex: int = 0, requiredPValue: float = 0.02, xclbinSrc: str = "", deviceIndices=None,
                 deferredStart: bool = False):
        """deviceIndices (an addition): several GPUs behind one object, one column shard each.
        deferredStart (an addition): Havac::DeferredStart -- the device layer starts under the caller's next calls."""
        self._L = load_host()
        h = C.c_void_p()
        if deferredStart:
            rc = self._L.havac_host_create_deferred(deviceIndex, requiredPValue, C.byref(h))
        elif deviceIndices is not None:
            arr = (C.c_uint32 * len(deviceIndices))(*deviceIndices)
            rc = self._L.havac_host_create_multi(arr, len(deviceIndices), requiredPValue, C.byref(h))
        else:
            rc = self._L.havac_host_create(deviceIndex, requiredPValue, C.byref(h))
        if rc != 0:
            raise_for(rc, "could not create Havac (no gfx950 device?)")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.havac_host_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise_for(rc, (self._L.havac_host_last_error(self._h) or b"").decode())
        return rc

    def loadSequence(self, fastaSrc: str):
        self._check(self._L.havac_host_load_sequence(self._h, os.fsencode(fastaSrc)))

    def loadPhmm(self, phmmSrc: str):
        self._check(self._L.havac_host_load_phmm(self._h, os.fsencode(phmmSrc)))

    def runHardwareClient(self):
        self._check(self._L.havac_host_run(self._h))

    def runHardwareClientAsync(self):
        self._check(self._L.havac_host_run_async(self._h))

    def waitHardwareClientAsync(self):
        self._check(self._L.havac_host_wait(self._h))

    def abortHardwareClient(self):
        self._check(self._L.havac_host_abort(self._h))

    def currentHardwareState(self) -> int:
        return self._check(self._L.havac_host_state(self._h))

    def setBoundaryMode(self, on: bool):
        """Not in the reference: score every (model, record) pair on its own (host/test/Ssv.cpp semantics)."""
        self._check(self._L.havac_host_set_boundary_mode(self._h, int(bool(on))))

    def setBothStrands(self, on: bool):
        """Not in the reference: also score the reverse complement of every record (nhmmer's default)."""
        self._check(self._L.havac_host_set_both_strands(self._h, int(bool(on))))

    def setDevicePacking(self, on: bool):
        """Not in the reference: pack the text on the GPU (default) or on the host."""
        self._check(self._L.havac_host_set_device_packing(self._h, int(bool(on))))

    def setHitCapacity(self, n: int):
        self._check(self._L.havac_host_set_hit_capacity(self._h, n))

    def setPipelineDepth(self, depth: int):
        """runs in flight (Havac::setPipelineDepth; 1 = the reference's one at a time)"""
        self._check(self._L.havac_host_set_pipeline_depth(self._h, depth))

    def getHitsFromFinishedRun(self):
        n = C.c_uint32(0)
        self._check(self._L.havac_host_get_hits(self._h, None, None, None, None, 0, C.byref(n)))
        sp = np.empty(n.value, np.uint64)
        si, pp, pi = (np.empty(n.value, np.uint32) for _ in range(3))
        self._check(self._L.havac_host_get_hits(self._h, sp.ctypes.data, si.ctypes.data, pp.ctypes.data,
                                                pi.ctypes.data, n.value, C.byref(n)))
        rev = np.zeros(n.value, np.uint8)
        self._check(self._L.havac_host_get_hit_strands(self._h, rev.ctypes.data, n.value, C.byref(n)))
        self._L.havac_host_next_run(self._h)      # (several runs open: the next call fetches the next run)
        return [HavacHit(int(a), int(b), int(c), int(d), bool(e)) for a, b, c, d, e in zip(sp, si, pp, pi, rev)]

    def getWindowsFromFinishedRun(self, flank: int = 0):
        """Not in the reference: the run's hits merged into windows (Havac.hpp: HavacWindow)."""
        n = C.c_uint32(0)
        self._check(self._L.havac_host_get_windows(self._h, flank, *([None] * 8), 0, C.byref(n)))
        arrays = _window_arrays(n.value)
        self._check(self._L.havac_host_get_windows(self._h, flank, *[a.ctypes.data for a in arrays], n.value, C.byref(n)))
        return _windows_from_arrays(arrays, n.value)

    def rawHits(self) -> np.ndarray:
        n = C.c_uint32(0)
        self._check(self._L.havac_host_get_raw_hits(self._h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint64)
        self._check(self._L.havac_host_get_raw_hits(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def lastRunMs(self):
        a, b = C.c_float(0), C.c_float(0)
        self._check(self._L.havac_host_last_run_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value


@dataclass(frozen=True)
class HavacWindow:   # Havac.hpp: HavacWindow (not in the reference)
    sequenceIndex: int
    phmmIndex: int
    reverseStrand: bool
    sequenceStart: int
    sequenceEnd: int
    phmmFirst: int
    phmmLast: int
    hitCount: int


def _window_arrays(n):
    return [np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.uint8), np.empty(n, np.uint64),
            np.empty(n, np.uint64), np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.uint32)]


def _windows_from_arrays(arrays, n):
    si, pi, rs, st, en, pf, pl, hc = arrays
    return [HavacWindow(int(si[i]), int(pi[i]), bool(rs[i]), int(st[i]), int(en[i]), int(pf[i]), int(pl[i]), int(hc[i]))
            for i in range(n)]


# ---- host-only stages (no device) --------------------------------------------

def pack_fasta(path: str, seed: int = -1):
    """FastaVector + SequencePreprocessor -> (packed uint8, nchars incl. terminators, nrecords)."""
    L = load_host()
    nb, nc, nr = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
    rc = L.havac_host_pack_fasta(os.fsencode(path), -1, None, 0, C.byref(nb), C.byref(nc), C.byref(nr))
    if rc != 0:
        raise_for(rc, f"could not read {path}")
    out = np.empty(nb.value, np.uint8)
    rc = L.havac_host_pack_fasta(os.fsencode(path), seed, out.ctypes.data, out.size, C.byref(nb), C.byref(nc), C.byref(nr))
    if rc != 0:
        raise_for(rc, f"could not read {path}")
    return out, nc.value, nr.value


def text_and_patches(path: str, seed: int = -1):
    """What loadSequence sends to the GPU when it packs there -> (chars uint8, patch columns uint64, patch symbols uint8)."""
    L = load_host()
    nc, npatch = C.c_uint64(0), C.c_uint64(0)
    rc = L.havac_host_text_and_patches(os.fsencode(path), -1, None, 0, C.byref(nc), None, None, 0, C.byref(npatch))
    if rc != 0:
        raise_for(rc, f"could not read {path}")
    # the counting call above drew from rand() as well: the caller's seed is applied by the second call
    chars = np.empty(nc.value, np.uint8)
    cols = np.empty(npatch.value, np.uint64)
    syms = np.empty(npatch.value, np.uint8)
    rc = L.havac_host_text_and_patches(os.fsencode(path), seed, chars.ctypes.data, chars.size, C.byref(nc),
                                       cols.ctypes.data, syms.ctypes.data, cols.size, C.byref(npatch))
    if rc != 0:
        raise_for(rc, f"could not read {path}")
    return chars, cols, syms


def pack_fasta_layout(path: str, boundary_mode: bool, both_strands: bool, seed: int = -1):
    """The host packer's other layouts -> (packed bytes, separator bitmap bytes (empty without boundary mode),
    forward columns (0 without both strands))."""
    L = load_host()
    nb, mb, nf = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    rc = L.havac_host_pack_fasta_layout(os.fsencode(path), seed, int(boundary_mode), int(both_strands), None, 0, C.byref(nb),
                                        None, 0, C.byref(mb), C.byref(nf))
    if rc != 0:
        raise_for(rc, f"could not read {path}")
    packed, mask = np.empty(nb.value, np.uint8), np.empty(mb.value, np.uint8)
    rc = L.havac_host_pack_fasta_layout(os.fsencode(path), seed, int(boundary_mode), int(both_strands),
                                        packed.ctypes.data if packed.size else None, packed.size, C.byref(nb),
                                        mask.ctypes.data if mask.size else None, mask.size, C.byref(mb), C.byref(nf))
    if rc != 0:
        raise_for(rc, f"could not read {path}")
    return packed, mask, nf.value


def project_hmm(path: str, p_value: float = 0.02):
    """P7HmmReader + PhmmPreprocessor -> (int8 [rows,4], model lengths)."""
    L = load_host()
    nb, nm = C.c_uint64(0), C.c_uint32(0)
    rc = L.havac_host_project_hmm(os.fsencode(path), p_value, None, 0, C.byref(nb), C.byref(nm), None, 0)
    if rc != 0:
        raise_for(rc, f"could not read {path}")
    out = np.empty(nb.value, np.int8)
    lens = np.empty(nm.value, np.uint32)
    rc = L.havac_host_project_hmm(os.fsencode(path), p_value, out.ctypes.data, out.size, C.byref(nb), C.byref(nm),
                                  lens.ctypes.data, lens.size)
    if rc != 0:
        raise_for(rc, f"could not read {path}")
    return out.reshape(-1, 4), lens


def read_hmm_emissions(path: str):
    """P7HmmReader alone -> (float32 [positions, 4] match-emission file values of all models back to back, model count)."""
    L = load_host()
    n, nm = C.c_uint64(0), C.c_uint32(0)
    rc = L.havac_host_read_hmm_emissions(os.fsencode(path), None, 0, C.byref(n), C.byref(nm))
    if rc != 0:
        raise_for(rc, f"could not read {path}")
    out = np.empty(n.value, np.float32)
    rc = L.havac_host_read_hmm_emissions(os.fsencode(path), out.ctypes.data, out.size, C.byref(n), C.byref(nm))
    if rc != 0:
        raise_for(rc, f"could not read {path}")
    return out.reshape(-1, 4), nm.value


def scaling_factor(mu, lam, max_length, model_length, p_value) -> float:
    return float(load_host().havac_host_scaling_factor(mu, lam, max_length, model_length, p_value))


def project_model(mu, lam, max_length, emissions, p_value) -> np.ndarray:
    """p7HmmProjectForThreshold256 on explicit parameters: emissions [L,4] float32 file values -> int8 [L,4]."""
    em = np.ascontiguousarray(emissions, dtype=np.float32)
    out = np.empty(em.shape, dtype=np.int8)
    rc = load_host().havac_host_project_model(mu, lam, int(max_length), em.shape[0], p_value, em.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise_for(rc, "bad argument to havac_host_project_model")
    return out


def gumbel_invsurv(p, mu, lam) -> float:
    return float(load_host().havac_host_gumbel_invsurv(p, mu, lam))


def project_score(emission_score, multiplier) -> float:
    return float(load_host().havac_host_project_score(emission_score, multiplier))


def resolve_hits(fasta_path: str, hmm_path: str, raw: np.ndarray):
    L = load_host()
    raw = np.ascontiguousarray(raw, np.uint64)
    n = C.c_uint32(0)
    sp = np.empty(raw.size, np.uint64)
    si, pp, pi = (np.empty(raw.size, np.uint32) for _ in range(3))
    rc = L.havac_host_resolve_hits(os.fsencode(fasta_path), os.fsencode(hmm_path), raw.ctypes.data, raw.size,
                                   sp.ctypes.data, si.ctypes.data, pp.ctypes.data, pi.ctypes.data, raw.size, C.byref(n))
    if rc != 0:
        raise_for(rc, "could not resolve hits")
    k = n.value
    return _hits_from_arrays(sp[:k], si[:k], pp[:k], pi[:k])


def merge_windows(hits, model_lengths, record_lengths, flank: int = 0):
    """havacMergeHitsToWindows: a list of HavacHit -> merged HavacWindow list."""
    L = load_host()
    sp = np.array([h.sequencePosition for h in hits], np.uint64)
    si = np.array([h.sequenceIndex for h in hits], np.uint32)
    pp = np.array([h.phmmPosition for h in hits], np.uint32)
    pi = np.array([h.phmmIndex for h in hits], np.uint32)
    rs = np.array([1 if h.reverseStrand else 0 for h in hits], np.uint8)
    ml = np.ascontiguousarray(model_lengths, np.uint32)
    rl = np.ascontiguousarray(record_lengths, np.uint64)
    arrays = _window_arrays(len(hits))
    n = C.c_uint32(0)
    rc = L.havac_host_merge_windows(sp.ctypes.data, si.ctypes.data, pp.ctypes.data, pi.ctypes.data, rs.ctypes.data,
                                    len(hits), ml.ctypes.data, ml.size, rl.ctypes.data, rl.size, flank,
                                    *[a.ctypes.data for a in arrays], len(hits), C.byref(n))
    if rc != 0:
        raise_for(rc, "could not merge windows")
    return _windows_from_arrays(arrays, n.value)
