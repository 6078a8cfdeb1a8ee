import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from havac_amd import synth
from havac_amd.dist import ShardedSsv
dev = torch.device("cuda", 0)
ncols = 100_012_032
d_seq = torch.from_numpy(synth.random_packed(ncols, synth.SEED_SEQUENCE)).to(dev)
eng = ShardedSsv(1 << 23, dev)
for nrows in (1024, 64):
    base, cons = synth.dfam_like_model(nrows, synth.SEED_MODEL)
    for name, every in (("all chunks safe", 0), ("one strongly negative stretch per 1024 rows", 1024), ("per 128 rows", 128), ("per 32 rows (every chunk unsafe)", 32)):
        model = base.copy()
        if every:
            for r in range(5, nrows, every):
                for k in range(3):
                    if r + k < nrows:
                        row = model[r + k]; best = row.argmax(); row[:] = -100; row[best] = 30
        d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
        (_, found), ms = eng.run_many(30, d_seq, ncols, d_phmm, nrows)
        (_, found), ms = eng.run_many(30, d_seq, ncols, d_phmm, nrows)
        k = float(np.mean([m[0] for m in ms]))
        print(f"rows {nrows}, {name}: kernel {k:.4f} ms = {ncols * nrows / k / 1e9:.1f} TCUPS, {found} hits", flush=True)
