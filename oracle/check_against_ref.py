"""`make -C oracle check`: our restatement vs the reference's softSsv on random inputs.

TEST INFRASTRUCTURE ONLY.  Needs oracle/_ref (build container)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as O          # noqa: E402
from havac_amd import synth               # noqa: E402


def main() -> int:
    if not O.ref_available():
        print("oracle/_ref not built (no /root/reference here): nothing to check")
        return 0
    rng = np.random.default_rng(7)
    bad = 0
    cases = [(1, 1), (1, 5), (5, 1), (3, 5), (100, 12288), (1024, 36864), (37, 1000), (20000, 3000)]
    for k in range(24):
        cases.append((int(rng.integers(1, 400)), int(rng.integers(1, 5000))))
    for nrows, n in cases:
        kind = int(rng.integers(0, 3))
        if kind == 0:
            model = rng.integers(-128, 128, size=(nrows, 4)).astype(np.int8)
        elif kind == 1:
            model, _ = synth.dfam_like_model(nrows, int(rng.integers(1 << 30)))
        else:
            model = rng.integers(60, 128, size=(nrows, 4)).astype(np.int8)   # dense hits
        sym = rng.integers(0, 4, size=n, dtype=np.uint8)
        a = O.ssv(sym, model)
        b = O.ssv_reference(sym, model)
        c = O.ssv_mt(sym, model, nthreads=3)
        ok = np.array_equal(a, b) and np.array_equal(a, c)
        lo, hi = sorted(int(x) for x in rng.integers(0, n + 1, size=2))
        w = O.ssv_window(sym, model, lo, hi)
        _, cols = O.unpack_hits(b)
        ok = ok and np.array_equal(w, b[(cols >= lo) & (cols < hi)])
        print(f"rows={nrows:6d} cols={n:6d} kind={kind} hits={a.size:8d} {'ok' if ok else 'MISMATCH'}")
        bad += not ok
    # speed, for the record
    model, _ = synth.dfam_like_model(1024)
    sym = synth.random_symbols(200_000, pad=False)
    for name, fn in (("restatement", O.ssv), ("reference", O.ssv_reference)):
        t = time.time(); h = fn(sym, model); dt = time.time() - t
        print(f"{name}: 1024 x 200000 in {dt:.2f}s = {1024*200000/dt/1e9:.3f} GCUPS, {h.size} hits")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
