"""Config C4 at full height on one GPU: the shard rank 5 of an 8-GPU node computes -- 1 Gbp database, the 1000-model
collection (~5e5 rows), 1.25e8 columns, ~6.3e13 cells -- timed, and every record of a 1.5e7-column stretch of it compared
with the CPU checker's vectorised route.   python tools/c4_shard_check.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from havac_amd import synth  # noqa: E402
from havac_amd.ssv import SsvContext, shard_columns  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

dev = torch.device("cuda", 0)
n = 81_381 * synth.SEGMENT                                  # 1,000,009,728 columns
model, cons = synth.model_collection(synth.model_lengths(1000), 2101)
nrows = model.shape[0]
packed = synth.random_packed(n, 777)
lo, hi = shard_columns(n, 5, 8)
print(f"shard 5 of 8: columns [{lo}, {hi}), {nrows} rows, {(hi - lo) * nrows:.3g} cells", flush=True)
d_seq = torch.from_numpy(packed).to(dev)
d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
cap = 1 << 30
hits = torch.empty(cap, dtype=torch.int64, device=dev)
ctx = SsvContext()
stream = torch.cuda.current_stream(dev).cuda_stream
for rep in range(2):
    t0 = time.perf_counter()
    ctx.enqueue(d_seq.data_ptr(), n, d_phmm.data_ptr(), nrows, hits.data_ptr(), cap, 5, 8, 0, stream)
    found = ctx.finish()
    wall = time.perf_counter() - t0
    k, tot = ctx.last_ms()
    print(f"run {rep}: {found} records, kernel {k:.1f} ms = {(hi - lo) * nrows / k / 1e9:.1f} TCUPS, with ordering {tot:.1f} ms, "
          f"wall {wall * 1e3:.1f} ms", flush=True)
# a stretch in the middle of the shard, and the one at its left edge (where the halo ends)
for a in (lo, (lo + hi) // 2 // 4 * 4):
    b = a + 15_000_000
    rec = hits[:found]
    cols = ((rec >> 14) & 0x3FFFFFF) * synth.SEGMENT + (rec & 0x3FFF)
    mine = rec[(cols >= a) & (cols < b)].cpu().numpy().view(np.uint64)
    start = max(0, a - (nrows - 1)) // 4 * 4
    sym = synth.unpack_2bit(packed[start // 4: b // 4])
    t0 = time.perf_counter()
    want = O.ssv_fast(sym, model, nthreads=16, cap=mine.size + (1 << 24))
    rows_w, cols_w = O.unpack_hits(want)
    keep = cols_w + np.uint64(start) >= np.uint64(a)
    want = O.device_order(O.pack_hits(rows_w[keep], cols_w[keep] + np.uint64(start)))
    print(f"columns [{a}, {b}): {mine.size} records on the GPU, {want.size} from the oracle ({time.perf_counter() - t0:.0f} s), "
          f"identical: {bool(np.array_equal(mine, want))}", flush=True)
ctx.close()
