"""The program tools/pmc_passes.sh puts behind `rocprofv3 --pmc ... --`: a few launches of the SSV kernel through the handle
API (ctypes + numpy only, nothing is re-executed).   python3 tools/pmc_probe.py [c2|c3|c5] [dfam|nohit] [launches]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from havac_amd import synth  # noqa: E402
from havac_amd.hw_client import HavacHwClient  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
kind = sys.argv[2] if len(sys.argv) > 2 else "dfam"
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if workload == "c3":
    model, cons = synth.model_collection(synth.model_lengths(1000), synth.SEED_MODEL)
    ncols = 10_002_432
elif workload == "c5":
    model, cons = synth.dfam_like_model(20000, synth.SEED_MODEL)
    ncols = 100_012_032
else:
    model, cons = synth.dfam_like_model(1024, synth.SEED_MODEL)
    ncols = 100_012_032
if kind == "nohit":
    model = np.full(model.shape, -40, np.int8)
packed = synth.random_packed(ncols, synth.SEED_SEQUENCE)
c = HavacHwClient()
c.setHitCapacity(max(1 << 20, int(ncols * model.shape[0] * 4e-5)))
c.writeSequence(packed)
c.writePhmm(model)
for _ in range(launches):
    c.invokeHavacSsvAsync()
    c.waitForHavacSsvAsync()
print(workload, kind, "hits", c.getNumHits(), "kernel ms", c.lastRunMs()[0], flush=True)
c.close()
