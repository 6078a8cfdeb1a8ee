"""One run at a time through the handle API (the reference's API path), the run split into 1, 2, 3 or 4 column shards that are
all on the SAME GPU (havac_dev_create_multi with one device named several times): do the shards' kernels and orderings overlap
to the run's advantage?   python tools/split_run_probe.py [rows ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from havac_amd import synth  # noqa: E402
from havac_amd.hw_client import HavacHwClient  # noqa: E402

ncols = 100_012_032
packed = synth.random_packed(ncols, synth.SEED_SEQUENCE)
for nrows in [int(a) for a in sys.argv[1:]] or [1024, 64]:
    model, _ = synth.dfam_like_model(nrows, synth.SEED_MODEL)
    for parts in (1, 2, 3, 4):
        c = HavacHwClient(deviceIndices=[0] * parts)
        c.setHitCapacity(4 << 20)
        c.writeSequence(packed)
        c.writePhmm(model)
        for _ in range(30):
            c.invokeHavacSsvAsync(); c.waitForHavacSsvAsync()
        t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            c.invokeHavacSsvAsync(); c.waitForHavacSsvAsync()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"rows {nrows}, the run as {parts} shard(s) on one GPU: {ms:.4f} ms per run = {ncols * nrows / ms / 1e9:.1f} TCUPS, {c.getNumHits()} hits", flush=True)
        c.close()
