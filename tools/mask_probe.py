"""Kernel time of C2 with and without a separator mask (boundary mode: what the file-level API always sets when a FASTA has more than
one record).   python tools/mask_probe.py [separators ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from havac_amd import synth  # noqa: E402
from havac_amd.hw_client import HavacHwClient  # noqa: E402

ncols = 100_012_032
for nrows in (1024, 64):
    model, _ = synth.dfam_like_model(nrows, synth.SEED_MODEL)
    packed = synth.random_packed(ncols, synth.SEED_SEQUENCE)
    c = HavacHwClient()
    c.setHitCapacity(4 << 20)
    for nsep in [0] + ([int(a) for a in sys.argv[1:]] or [100, 10000]):
        c.writeSequence(packed)
        if nsep:
            rng = np.random.default_rng(5)
            seps = np.unique(rng.integers(0, ncols // 2, size=nsep) * 2)
            mask = np.zeros(ncols // 16, np.uint8)
            np.bitwise_or.at(mask, seps // 16, (1 << ((seps // 2) % 8)).astype(np.uint8))
            c.writeSeparatorMask(mask)
        c.writePhmm(model)
        ks = []
        for rep in range(25):
            c.invokeHavacSsvAsync()
            c.waitForHavacSsvAsync()
            if rep >= 5:
                ks.append(c.lastRunMs()[0])
        print(f"rows {nrows}, {nsep} separators: kernel {np.mean(ks):.4f} ms = {ncols * nrows / np.mean(ks) / 1e9:.1f} TCUPS, {c.getNumHits()} hits", flush=True)
    c.close()
