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
        # the bare context on a stream of the caller's (no pipe): what the pipe's bookkeeping adds to a strictly serial step
        from havac_amd.ssv import SsvContext
        ctx = SsvContext()
        hits = torch.zeros(cap, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(dev)
        for name, stream in (("torch's current (null) stream", torch.cuda.current_stream(dev).cuda_stream), ("a stream of its own", side.cuda_stream)):
            for _ in range(20):
                ctx.enqueue(d_seq.data_ptr(), ncols, d_phmm.data_ptr(), rows, hits.data_ptr(), hits.numel(), 0, 1, 0, stream)
                found = ctx.finish()
            t0 = time.perf_counter()
            for _ in range(60):
                ctx.enqueue(d_seq.data_ptr(), ncols, d_phmm.data_ptr(), rows, hits.data_ptr(), hits.numel(), 0, 1, 0, stream)
                found = ctx.finish()
            step = (time.perf_counter() - t0) / 60 * 1e3
            print(f"rows {rows} bare context on {name}: found {found} step {step:.4f} ms = {ncols * rows / step / 1e9:.2f} TCUPS", flush=True)
        ctx.close()
        del hits
        for depth, streams, order_streams in ((1, None, 0), (2, 2, 0), (3, 2, 0), (3, 3, 0), (2, 1, 1), (3, 1, 1), (2, 2, 1), (3, 2, 1)):
            eng = ShardedSsv(cap, dev, depth=depth, kernel_streams=streams)
            keep = []
            if order_streams:      # every slot orders on a HIGH-priority stream of its own (round 4 had them at low priority)
                import ctypes as C
                for i in range(depth):
                    st = torch.cuda.Stream(dev, priority=-1)
                    keep.append(st)
                    eng._L.havac_ssv_set_order_stream(C.c_void_p(eng._L.havac_pipe_context(eng._h, i)), C.c_void_p(st.cuda_stream))
            eng.run_many(20, d_seq, ncols, d_phmm, rows)
            torch.cuda.synchronize(dev)
            n = 120
            t0 = time.perf_counter()
            (recs, found), ms = eng.run_many(n, d_seq, ncols, d_phmm, rows, inputs_ready=True)
            torch.cuda.synchronize(dev)
            step = (time.perf_counter() - t0) / n * 1e3
            k = sum(m[0] for m in ms) / len(ms)
            tot = sum(m[1] for m in ms) / len(ms)
            print(f"rows {rows} depth {depth} kernel streams {streams}{' + an ordering stream per slot' if order_streams else ''}: found {found} step {step:.4f} ms = {ncols * rows / step / 1e9:.2f} TCUPS; kernel {k:.4f} ms, enqueue-to-ordered {tot:.4f} ms", flush=True)
            eng.release()


if __name__ == "__main__":
    main()
