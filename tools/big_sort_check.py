"""havac_ssv_sort_hits on more than 2^32 records: is the output in device order, and is it the same multiset (sum and
xor of all records)?  No torch (its own kernels mis-index tensors of more than 2^32 elements on this build: an arange of
4.5e9 int64 reads 0 from element 4,026,531,840 on): HIP through ctypes, numpy on the host.
python tools/big_sort_check.py [count]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from havac_amd import _lib  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_500_000_000
L = _lib.load()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipFree.argtypes = [C.c_void_p]
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), n * 8) == 0
rng = np.random.default_rng(5)
chunk = 1 << 27
total_sum = total_xor = 0
for a in range(0, n, chunk):
    m = min(chunk, n - a)
    rec = (rng.integers(0, 12_288, m, dtype=np.uint64) | (rng.integers(0, 81_381, m, dtype=np.uint64) << np.uint64(14))
           | (rng.integers(0, 503_329, m, dtype=np.uint64) << np.uint64(40)))
    total_sum = (total_sum + int(rec.sum(dtype=np.uint64))) & ((1 << 64) - 1)
    total_xor ^= int(np.bitwise_xor.reduce(rec))
    assert hip.hipMemcpy(C.c_void_p(d.value + a * 8), rec.ctypes.data, m * 8, 1) == 0
ctx = C.c_void_p()
assert L.havac_ssv_ctx_create(C.byref(ctx)) == 0
rc = L.havac_ssv_sort_hits(ctx, d, n, None)
assert rc == 0, (rc, L.havac_ssv_ctx_last_error(ctx))
assert hip.hipDeviceSynchronize() == 0
bad = 0
s2 = x2 = 0
last = None
buf = np.empty(chunk, np.uint64)
for a in range(0, n, chunk):
    m = min(chunk, n - a)
    assert hip.hipMemcpy(buf.ctypes.data, C.c_void_p(d.value + a * 8), m * 8, 2) == 0
    rec = buf[:m]
    s2 = (s2 + int(rec.sum(dtype=np.uint64))) & ((1 << 64) - 1)
    x2 ^= int(np.bitwise_xor.reduce(rec))
    key = (((rec >> np.uint64(14)) & np.uint64(0x3FFFFFF)) << np.uint64(38)) | ((rec >> np.uint64(40)) << np.uint64(14)) | (rec & np.uint64(0x3FFF))
    bad += int((key[1:] < key[:-1]).sum())
    if last is not None and key[0] < last:
        bad += 1
    last = key[-1]
print(f"{n} records: same multiset {(s2, x2) == (total_sum, total_xor)}, out-of-order neighbours {bad}", flush=True)
L.havac_ssv_ctx_destroy(ctx)
hip.hipFree(d)
