"""Steps per second of the whole pass (pad, SSV kernel, ordering) with 1 and 2 passes in flight.
   python tools/pipeline_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from havac_amd import synth  # noqa: E402
from havac_amd.dist import ShardedSsv  # noqa: E402

dev = torch.device("cuda", 0)
ncols, nrows = 100_012_032, 1024
model, cons = synth.dfam_like_model(nrows, synth.SEED_MODEL)
packed = synth.random_packed(ncols, synth.SEED_SEQUENCE)
d_seq = torch.from_numpy(packed).to(dev)
d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
steps = 40
ref = None
for depth in (1, 2, 3):
    eng = ShardedSsv(1 << 22, dev, depth=depth)
    for _ in range(3):
        eng.run(d_seq, ncols, d_phmm, nrows)
    torch.cuda.synchronize()
    kms = []
    t0 = time.perf_counter()
    for k in range(steps):
        eng.submit(d_seq, ncols, d_phmm, nrows)
        if len(eng.in_flight) == depth:
            hits, found = eng.collect()
            kms.append(eng.ctx.last_ms()[0])
    while eng.in_flight:
        hits, found = eng.collect()
        kms.append(eng.ctx.last_ms()[0])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    got = hits.cpu().numpy().copy()
    if ref is None:
        ref = got
    print(f"depth {depth}: {dt * 1e3:.4f} ms/step = {ncols * nrows / dt / 1e12:.2f} TCUPS, "
          f"kernel avg {np.mean(kms):.4f} ms, hits {found}, same list {np.array_equal(got, ref)}", flush=True)
    eng.close()
