"""Kernel time against model height (32-row chunks per tile) on 100 Mbp, with a model that cannot hit and with the
Dfam-like one.   python tools/rows_probe.py [--tuning=a,b,...] [--columns=N] [rows ...]     (tuning: ShardedSsv's, see bench.py --tuning;
--columns: the sequence's length, rounded up to whole segments -- what a launch's start and end cost shows as the rate's dependence on N)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from havac_amd import synth  # noqa: E402
from havac_amd.dist import ShardedSsv  # noqa: E402

dev = torch.device("cuda", 0)
ncols = 100_012_032
args = sys.argv[1:]
tuning = None
while args and args[0].startswith("--"):
    if args[0].startswith("--tuning="):
        tuning = [int(v) for v in args.pop(0)[len("--tuning="):].split(",")]
    elif args[0].startswith("--columns="):
        ncols = -(-int(float(args.pop(0)[len("--columns="):])) // 12288) * 12288
    else:
        raise SystemExit(f"unknown option {args[0]}")
packed = synth.random_packed(ncols, synth.SEED_SEQUENCE)
d_seq = torch.from_numpy(packed).to(dev)
eng = ShardedSsv(max(1 << 23, int(ncols * 1024 * 4e-5)), dev, tuning=tuning)
print(f"{ncols} columns", flush=True)
heights = [int(a) for a in args] or [32, 64, 96, 128, 160, 192, 256, 384, 512, 1024]
for nrows in heights:
    out = []
    for kind in ("nohit", "dfam"):
        model = synth.dfam_like_model(nrows, synth.SEED_MODEL)[0] if kind == "dfam" else np.full((nrows, 4), -3, np.int8)
        d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
        for _ in range(3):
            eng.run(d_seq, ncols, d_phmm, nrows)
        ms = []
        for _ in range(10):
            hits, found = eng.run(d_seq, ncols, d_phmm, nrows)
            ms.append(eng.ctx.last_ms()[0])
        out.append(f"{kind} {np.mean(ms):.4f} ms {ncols * nrows / np.mean(ms) / 1e9:.1f} TCUPS hits {found}")
    print(nrows, " | ".join(out), flush=True)
