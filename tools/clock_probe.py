"""Where the time of a tile goes, by in-kernel clocks: runs a build of libhavac_dev.so instrumented with s_memtime stamps
(made in the build container from a copy of ssv_kernels.hip.h with stamp() calls at the phase boundaries; never part of the
product) and prints the average cycles per wave between stamps.   python3 tools/clock_probe.py <instrumented .so> [rows ...]"""
import ctypes as C
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_path = os.path.join(ROOT, "havac_amd", "libhavac_dev.so")
kept = lib_path + ".kept"
shutil.copyfile(lib_path, kept)
shutil.copyfile(sys.argv[1], lib_path)
try:
    from havac_amd import synth, _lib                      # noqa: E402
    from havac_amd.hw_client import HavacHwClient          # noqa: E402
    L = C.CDLL(lib_path)
    L.havac_debug_clocks.argtypes = [C.c_void_p, C.c_int]
    names = ["waves", "block start -> items (lane words, outside entries)", "-> tile start (ticket, item decode)",
             "-> first symbols loaded and prepared", "-> chunk loop entry (window expanded, rows + second symbols issued)",
             "-> chunk loop done", "-> tile done (step behind the last chunk / hand-off)", "-> block end reached",
             "-> barrier passed"]
    ncols = 100_012_032
    packed = synth.random_packed(ncols, synth.SEED_SEQUENCE)
    c = HavacHwClient()
    c.writeSequence(packed)
    for nrows in [int(a) for a in sys.argv[2:]] or [32, 64, 1024]:
        for kind in ("nohit", "dfam"):
            model = synth.dfam_like_model(nrows, synth.SEED_MODEL)[0] if kind == "dfam" else np.full((nrows, 4), -3, np.int8)
            c.writePhmm(model)
            c.setHitCapacity(1 << 23)
            for _ in range(3):
                c.invokeHavacSsvAsync(); c.waitForHavacSsvAsync()
            buf = np.zeros(65536 * 8, np.uint32)
            L.havac_debug_clocks(buf.ctypes.data, 1)
            ms = []
            for _ in range(5):
                c.invokeHavacSsvAsync(); c.waitForHavacSsvAsync()
                ms.append(c.lastRunMs()[0])
            L.havac_debug_clocks(buf.ctypes.data, 1)
            d = buf.reshape(65536, 8).astype(np.float64)
            ran = d[:, 4] > 0                                   # waves that had a tile (the last launch's stamps)
            print(f"rows {nrows} {kind}: kernel {np.mean(ms) * 1e3:.1f} us, {int(ran.sum())} waves with a tile; s_memtime ticks per wave, mean (median):", flush=True)
            for k in range(8):
                print(f"    {names[k + 1]:75s} {d[ran, k].mean():10.1f} ({np.median(d[ran, k]):8.1f})")
            print(f"    {'sum':75s} {d[ran].sum(axis=1).mean():10.1f}")
    c.close()
finally:
    shutil.copyfile(kept, lib_path)
    os.remove(kept)
