import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from havac_amd import synth
from havac_amd.dist import ShardedSsv
dev = torch.device("cuda", 0)
ncols, nrows = 100_012_032, 1024
packed = synth.random_packed(ncols, 1001)
d_seq = torch.from_numpy(packed).to(dev)
for name, model in (("dfam", synth.dfam_like_model(nrows, 2001)[0]), ("nohit", np.full((nrows, 4), -40, np.int8)),
                    ("fewhit", np.where(np.random.default_rng(1).random((nrows, 4)) < 0.22, 30, -60).astype(np.int8))):
    d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
    eng = ShardedSsv(1 << 23, dev)
    for _ in range(5): eng.run(d_seq, ncols, d_phmm, nrows)
    ks = []
    for _ in range(20):
        m, found = eng.run(d_seq, ncols, d_phmm, nrows); ks.append(eng.ctx.last_ms()[0])
    print(name, "hits", found, "kernel ms", round(float(np.mean(ks)), 4), "TCUPS", round(ncols * nrows / np.mean(ks) / 1e9, 2), flush=True)
