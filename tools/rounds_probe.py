"""Is C2's no-hit rate held down by its half-empty last round of tiles?  Kernel rate at database sizes that make
9.54 (C2), 9.99, 10.01 and 19.98 rounds of 5120 wave slots.   python tools/rounds_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from havac_amd import synth
from havac_amd.dist import ShardedSsv
dev = torch.device("cuda", 0)
nrows = 1024
model = np.full((nrows, 4), -40, np.int8)
d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
for nseg in (8139, 8530, 8533, 8536, 17063):
    ncols = nseg * synth.SEGMENT
    tiles = (ncols + nrows + 2047) // 2048
    d_seq = torch.from_numpy(synth.random_packed(ncols, 1001)).to(dev)
    eng = ShardedSsv(1 << 20, dev)
    for _ in range(3): eng.run(d_seq, ncols, d_phmm, nrows)
    ks = []
    for _ in range(10):
        eng.run(d_seq, ncols, d_phmm, nrows); ks.append(eng.ctx.last_ms()[0])
    print(f"{ncols} columns, {tiles} tiles = {tiles / 5120:.2f} rounds: {np.mean(ks):.4f} ms, {ncols * nrows / np.mean(ks) / 1e9:.2f} TCUPS", flush=True)
    eng.close(); del d_seq
