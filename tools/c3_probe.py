"""Kernel time on config C3 as BASELINE.json states it (1000 models, lengths log-uniform in 50..2000, x 10 Mbp) and on
the same model collection against 100 Mbp.   python tools/c3_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from havac_amd import synth  # noqa: E402
from havac_amd.dist import ShardedSsv  # noqa: E402

dev = torch.device("cuda", 0)
model, cons = synth.model_collection(synth.model_lengths(1000), 2101)
d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
nrows = model.shape[0]
for nseg in (814, 8139):
    ncols = nseg * synth.SEGMENT
    d_seq = torch.from_numpy(synth.random_packed(ncols, 1303)).to(dev)
    eng = ShardedSsv(1 << 29, dev)
    ms, tot = [], []
    for k in range(4):
        hits, found = eng.run(d_seq, ncols, d_phmm, nrows)
        if k:
            a, b = eng.ctx.last_ms()
            ms.append(a)
            tot.append(b)
    print(f"{nrows} rows x {ncols} columns: kernel {np.mean(ms):.2f} ms = {ncols * nrows / np.mean(ms) / 1e9:.1f} TCUPS, "
          f"with ordering {np.mean(tot):.2f} ms = {ncols * nrows / np.mean(tot) / 1e9:.1f} TCUPS, {found} hits, "
          f"{(ncols + nrows + 2047) // 2048} tiles", flush=True)
    eng.close()
    del d_seq
