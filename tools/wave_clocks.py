"""When the waves of the resident-table kernel start and end (experiments: a build with -DHAVAC_WAVE_CLOCKS as
tools/_bin/ab/libW.so).   python3 tools/wave_clocks.py [rows ...]"""
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
shutil.copyfile(os.path.join(ROOT, "tools", "_bin", "ab", "libW.so"), lib_path)
try:
    import torch
    from havac_amd import synth
    from havac_amd.dist import ShardedSsv
    L = C.CDLL(lib_path)
    dev = torch.device("cuda", 0)
    ncols = 100_012_032
    d_seq = torch.from_numpy(synth.random_packed(ncols, synth.SEED_SEQUENCE)).to(dev)
    eng = ShardedSsv(1 << 23, dev)
    for nrows in [int(a) for a in sys.argv[1:]] or [32, 256]:
        model = np.full((nrows, 4), -3, np.int8)
        d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
        for _ in range(4):
            eng.run(d_seq, ncols, d_phmm, nrows)
        ms = eng.ctx.last_ms()[0]
        n = 16384
        buf = np.zeros(n * 4, np.uint64)
        assert L.havac_debug_wave_clocks(buf.ctypes.data_as(C.c_void_p), n) == 0
        d = buf.reshape(n, 4)
        t0 = d[:, 0].min()
        start = (d[:, 0] - t0).astype(np.float64) / 100.0          # us (100 MHz)
        end = (d[:, 1] - t0).astype(np.float64) / 100.0
        hw = d[:, 2]
        simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7          # gfx9 HW_ID layout
        print(f"rows {nrows}: kernel {ms * 1e3:.1f} us; wave start us: min {start.min():.1f} median {np.median(start):.1f} max {start.max():.1f}; "
              f"wave end us: min {end.min():.1f} p10 {np.percentile(end, 10):.1f} median {np.median(end):.1f} p90 {np.percentile(end, 90):.1f} max {end.max():.1f}", flush=True)
        print("   run time us by tiles walked:", {int(k): round(float((end - start)[d[:, 3] == k].mean()), 1) for k in np.unique(d[:, 3])})
        # the waves that share a SIMD: how far apart do they finish?
        key = (se.astype(np.int64) << 20) | (sh.astype(np.int64) << 16) | (cu.astype(np.int64) << 8) | simd.astype(np.int64)
        spread = []
        counts = []
        for k in np.unique(key):
            e = np.sort(end[key == k])
            counts.append(e.size)
            spread.append(e[-1] - e[0])
        print(f"   distinct (se, sh, cu, simd) keys {len(spread)}, waves per key min {min(counts)} max {max(counts)}; spread of end times within a key: mean {np.mean(spread):.1f} us, max {np.max(spread):.1f} us")
        hist, edges = np.histogram(end, bins=12)
        print("   histogram of wave end times (us):", [(round(float(a), 0), int(h)) for a, h in zip(edges[:-1], hist)])
finally:
    shutil.copyfile(kept, lib_path)
    os.remove(kept)
