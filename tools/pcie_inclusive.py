"""PCIe-inclusive rate of the handle API on config C2: host buffers in, host hit list out.

Never the bench `value` (that one has inputs resident in HBM); recorded in DESIGN.md for reference.
    python tools/pcie_inclusive.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (first: see tests/conftest.py)
from havac_amd import synth  # noqa: E402
from havac_amd.hw_client import HavacHwClient  # noqa: E402

model, _ = synth.dfam_like_model(1024, synth.SEED_MODEL)
packed = synth.random_packed(100_012_032, synth.SEED_SEQUENCE)
c = HavacHwClient()
c.setHitCapacity(4 << 20)
for rep in range(4):
    t0 = time.perf_counter()
    c.writeSequence(packed)
    c.writePhmm(model)
    t1 = time.perf_counter()
    c.invokeHavacSsvAsync()
    c.waitForHavacSsvAsync()
    t2 = time.perf_counter()
    hits = c.getHitList()
    t3 = time.perf_counter()
    cells = 100_012_032 * 1024
    print(f"rep {rep}: H2D {1e3*(t1-t0):.2f} ms, run {1e3*(t2-t1):.2f} ms, D2H {1e3*(t3-t2):.2f} ms ({hits.size} hits); "
          f"PCIe-inclusive {cells/(t3-t0)/1e9:.0f} GCUPS, run-only {cells/(t2-t1)/1e9:.0f} GCUPS", flush=True)
