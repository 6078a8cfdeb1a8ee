"""The program tools/pmc_passes.sh puts behind `rocprofv3 --pmc ... --`: a few launches of the SSV kernel through the handle
API (ctypes + numpy only, nothing is re-executed).   python3 tools/pmc_probe.py [c2|c3|c5|r<rows>] [dfam|nohit] [launches] [tuning]"""
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
tuning = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 and sys.argv[4] else []      # havac_dev_set_tuning
if workload == "c3":
    model, cons = synth.model_collection(synth.model_lengths(1000), synth.SEED_MODEL)
    ncols = 10_002_432
elif workload == "c5":
    model, cons = synth.dfam_like_model(20000, synth.SEED_MODEL)
    ncols = 100_012_032
elif workload.startswith("r"):                   # r<rows>: one model of that many rows x 100 Mbp
    model, cons = synth.dfam_like_model(int(workload[1:]), synth.SEED_MODEL)
    ncols = 100_012_032
else:
    model, cons = synth.dfam_like_model(1024, synth.SEED_MODEL)
    ncols = 100_012_032
if kind == "nohit":
    model = np.full(model.shape, -40, np.int8)
packed = synth.random_packed(ncols, synth.SEED_SEQUENCE)
c = HavacHwClient()
c.setHitCapacity(max(1 << 20, int(ncols * model.shape[0] * 4e-5)))
if tuning:
    c.setTuning(*tuning)
c.writeSequence(packed)
c.writePhmm(model)
ms = []
for _ in range(launches):
    c.invokeHavacSsvAsync()
    c.waitForHavacSsvAsync()
    ms.append(c.lastRunMs()[0])
later = ms[1:] or ms
print(workload, kind, "hits", c.getNumHits(), "kernel ms", ms[-1], "mean of the launches after the first", round(sum(later) / len(later), 4),
      "min", round(min(ms), 4), flush=True)
c.close()
