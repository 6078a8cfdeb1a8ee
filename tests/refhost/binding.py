"""ctypes view of tests/_refhost/libhavac_refhost.so: the REFERENCE's own `Havac` class (host/Havac.cpp, compiled from
where it lies by tests/refhost/Makefile in the build container) running over this repository's C ABI through
integration/HavacHwClient.hpp.  TEST INFRASTRUCTURE ONLY -- nothing under havac_amd/ or in bench.py imports this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "_refhost")
REFHOST_LIB = os.path.join(OUT, "libhavac_refhost.so")
SSV_REF2_LIB = os.path.join(OUT, "libssv_ref2.so")
REFERENCE = "/root/reference"

_vp = C.c_void_p
SIGNATURES = {
    "refhost_create": (C.c_int, [C.c_uint32, C.c_float, C.POINTER(C.c_void_p)]),
    "refhost_destroy": (None, [_vp, C.c_int]),
    "refhost_load_phmm": (C.c_int, [_vp, C.c_char_p]),
    "refhost_load_sequence": (C.c_int, [_vp, C.c_char_p]),
    "refhost_run": (C.c_int, [_vp]),
    "refhost_run_async": (C.c_int, [_vp]),
    "refhost_wait": (C.c_int, [_vp]),
    "refhost_abort": (C.c_int, [_vp]),
    "refhost_state": (C.c_int, [_vp]),
    "refhost_get_hits": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "refhost_hit_to_string": (C.c_int, [_vp, C.c_uint32, C.c_char_p, C.c_uint32]),
    "refhost_last_error": (C.c_char_p, [_vp]),
    "refhost_srand": (None, [C.c_uint]),
}


def build_if_possible() -> bool:
    """In the build container (the reference tree is there) run the committed recipe; on the GPU box use what travelled."""
    if os.path.isfile(os.path.join(REFERENCE, "host", "Havac.cpp")):
        r = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("tests/refhost/Makefile failed:\n" + r.stdout + r.stderr)
    return os.path.isfile(REFHOST_LIB)


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        lib = C.CDLL(REFHOST_LIB)          # RTLD_LOCAL: its Havac never meets libhavac.so's
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class RefHostError(Exception):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code      # -1 length_error, -2 logic_error, -3 runtime_error, -4 bad_alloc (include/havac_dev.h)


class ReferenceHavac:
    """The reference's Havac, method for method (host/Havac.hpp:42-107)."""

    def __init__(self, deviceIndex=0, requiredPValue=0.02):
        self._L = load()
        h = C.c_void_p()
        rc = self._L.refhost_create(deviceIndex, requiredPValue, C.byref(h))
        if rc != 0:
            raise RefHostError(rc, "the reference's Havac constructor threw")
        self._h = h
        self._phmm_loaded = False

    def _check(self, rc):
        if rc < 0:
            raise RefHostError(rc, (self._L.refhost_last_error(self._h) or b"").decode())
        return rc

    def loadPhmm(self, path):
        self._phmm_loaded = True           # readP7Hmm has initialised the list whatever it returns
        self._check(self._L.refhost_load_phmm(self._h, os.fsencode(path)))

    def loadSequence(self, path):
        self._check(self._L.refhost_load_sequence(self._h, os.fsencode(path)))

    def runHardwareClient(self):
        self._check(self._L.refhost_run(self._h))

    def runHardwareClientAsync(self):
        self._check(self._L.refhost_run_async(self._h))

    def waitHardwareClientAsync(self):
        self._check(self._L.refhost_wait(self._h))

    def abortHardwareClient(self):
        self._check(self._L.refhost_abort(self._h))

    def currentHardwareState(self):
        return self._check(self._L.refhost_state(self._h))

    def getHitsFromFinishedRun(self):
        """-> list of (sequencePosition, sequenceIndex, phmmPosition, phmmIndex) in the order the reference returns them"""
        n = C.c_uint32(0)
        self._check(self._L.refhost_get_hits(self._h, None, None, None, None, 0, C.byref(n)))
        sp = np.empty(n.value, np.uint64)
        si, pp, pi = (np.empty(n.value, np.uint32) for _ in range(3))
        self._check(self._L.refhost_get_hits(self._h, sp.ctypes.data, si.ctypes.data, pp.ctypes.data, pi.ctypes.data,
                                             n.value, C.byref(n)))
        return list(zip(sp.tolist(), si.tolist(), pp.tolist(), pi.tolist()))

    def hitToString(self, i):
        buf = C.create_string_buffer(256)
        self._check(self._L.refhost_hit_to_string(self._h, i, buf, 256))
        return buf.value.decode()

    def close(self):
        if getattr(self, "_h", None):
            self._L.refhost_destroy(self._h, int(self._phmm_loaded))
            self._h = None


def reference_ssv_hits(fasta_path, hmm_path, p_value=0.02):
    """HitsFromSsv (host/test/Ssv.cpp:8-68) -> uint32 [n,4] rows (sequenceNumber, phmmNumber, sequencePosition, phmmPosition)."""
    lib = C.CDLL(SSV_REF2_LIB)
    lib.ssv_ref2_hits.restype = C.c_int
    lib.ssv_ref2_hits.argtypes = [C.c_char_p, C.c_char_p, C.c_float, _vp, _vp, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]
    n = C.c_uint64(0)
    rc = lib.ssv_ref2_hits(os.fsencode(fasta_path), os.fsencode(hmm_path), p_value, None, None, None, None, 0, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"ssv_ref2_hits: could not read the inputs ({rc})")
    cols = [np.empty(n.value, np.uint32) for _ in range(4)]
    rc = lib.ssv_ref2_hits(os.fsencode(fasta_path), os.fsencode(hmm_path), p_value, *[c.ctypes.data for c in cols],
                           n.value, C.byref(n))
    assert rc == 0
    return np.stack(cols, axis=1) if n.value else np.empty((0, 4), np.uint32)
