"""The strictly serial step (enqueue + finish, one pass at a time: what a caller of the reference's one-run-at-a-time API sees) and
the pipelined step for a few model heights, without bench.py's checks.  python tools/step_probe.py [rows ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from havac_amd.dist import ShardedSsv  # noqa: E402
import bench  # noqa: E402


def main():
    rows_list = [int(v) for v in sys.argv[1:]] or [64, 1024]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for rows in rows_list:
        model, packed, ncols, _, _ = bench.make_inputs("c2", 1, rows, 0)
        d_seq = torch.from_numpy(packed).to(dev)
        d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
        cap = max(1 << 20, int(ncols * rows * 4e-5))
        for depth in (1, 2):
            eng = ShardedSsv(cap, dev, depth=depth)
            def run(n):
                res, ms = None, []
                for _ in range(n):
                    eng.submit(d_seq, ncols, d_phmm, rows)
                    if len(eng.in_flight) == len(eng.slots):
                        res = eng.collect(); ms.append(eng.ctx.last_ms())
                while eng.in_flight:
                    res = eng.collect(); ms.append(eng.ctx.last_ms())
                return res, ms
            run(20)
            torch.cuda.synchronize(dev)
            n = 60
            t0 = time.perf_counter()
            (recs, found), ms = run(n)
            torch.cuda.synchronize(dev)
            step = (time.perf_counter() - t0) / n * 1e3
            k = sum(m[0] for m in ms) / len(ms)
            tot = sum(m[1] for m in ms) / len(ms)
            print(f"rows {rows} depth {depth}: found {found} step {step:.4f} ms = {ncols * rows / step / 1e9:.2f} TCUPS; kernel {k:.4f} ms, enqueue-to-ordered {tot:.4f} ms", flush=True)
            eng.release()


if __name__ == "__main__":
    main()
