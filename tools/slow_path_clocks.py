"""How long a wave is away in the slow path, and the shader clock a launch really runs at (experiments: a build with
-DHAVAC_SLOW_CLOCKS=2 as tools/_bin/ab/libS.so -- s_memtime at the slow path's entry and exit, summed per launch; with =1 only the
clock: s_memtime against the 100 MHz s_memrealtime over every workgroup's life, the kernel at its usual speed).
python3 tools/slow_path_clocks.py [rows ...]"""
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
shutil.copyfile(os.path.join(ROOT, "tools", "_bin", "ab", "libS.so"), lib_path)
try:
    import torch
    from havac_amd import synth
    from havac_amd.dist import ShardedSsv
    L = C.CDLL(lib_path)
    dev = torch.device("cuda", 0)
    ncols = 100_012_032
    d_seq = torch.from_numpy(synth.random_packed(ncols, synth.SEED_SEQUENCE)).to(dev)
    eng = ShardedSsv(1 << 25, dev)
    for nrows in [int(a) for a in sys.argv[1:]] or [1024, 64]:
        model = synth.dfam_like_model(nrows, synth.SEED_MODEL)[0]
        d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
        # back to back, as bench.py's passes are: a lone pass after a pause runs at the clock the chip idles at (rows 1024: 2.12 GHz
        # and 2.08 ms where the twentieth pass of a row takes 1.85)
        passes = max(3, min(40, int(2e12 / (ncols * nrows))))
        eng.run_many(passes, d_seq, ncols, d_phmm, nrows)
        buf = (C.c_uint64 * 4)()
        assert L.havac_debug_slow_clocks(buf, 1) == 0
        (_, found), times = eng.run_many(passes, d_seq, ncols, d_phmm, nrows)
        ms = float(np.mean([t[0] for t in times]))
        assert L.havac_debug_slow_clocks(buf, 1) == 0
        buf[0] //= passes; buf[1] //= passes
        cycles, entries = int(buf[0]), int(buf[1])
        ghz = int(buf[2]) / max(int(buf[3]), 1) * 0.1      # s_memtime ticks per 100 MHz tick, over every workgroup's life
        print(f"rows {nrows}: the shader clock the launch's workgroups saw: {ghz:.3f} GHz", flush=True)
        chunks = ncols * nrows / (2048 * 32)
        print(f"rows {nrows}: kernel {ms:.4f} ms (this build: two atomics per entry), {found} hits, {entries} slow-path entries, "
              f"{cycles / max(entries, 1):.0f} shader cycles per entry; a wave's chunk takes {ms * 1e-3 * 2.4e9 * 1024 * 6 / chunks:.0f} cycles of its life "
              f"(six waves per SIMD) -> an entry = {cycles / max(entries, 1) / (ms * 1e-3 * 2.4e9 * 1024 * 6 / chunks) * 8:.2f} windows of it", flush=True)
finally:
    shutil.copyfile(kept, lib_path)
    os.remove(kept)
