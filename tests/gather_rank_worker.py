"""One rank of tests/test_gpu_gather_ranks.py: `python gather_rank_worker.py RANK WORLD DIR`.

Every rank shares GPU 0.  libhavac_dev.so's havac_gather_* (include/havac_dev.h, level 3) run exactly as on an 8-GPU node,
but bound to tests/native/librccl_standin.so instead of librccl.so.1 (havac_gather_use_library), because RCCL refuses two
ranks on one device.  The 128-byte id travels through a file in DIR.  Each scenario writes what this rank saw into
DIR/result_RANK.json; rank 0 compares gathered lists with the CPU checker.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  -- first: see tests/conftest.py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from havac_amd import _lib, synth  # noqa: E402
from havac_amd.dist import RcclGather  # noqa: E402
from havac_amd.hw_client import CollectiveTimeout, LengthError  # noqa: E402
from havac_amd.ssv import SsvContext, shard_columns  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

STANDIN = os.path.join(ROOT, "tests", "native", "librccl_standin.so")


def exchange_id(rank, directory, tag):
    path = os.path.join(directory, f"id_{tag}")
    if rank == 0:
        uid = RcclGather.unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(path + ".tmp", path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 60:
            raise RuntimeError("rank 0 never wrote the id")
        time.sleep(0.01)
    with open(path, "rb") as f:
        return f.read()


def file_barrier(rank, world, directory, tag):
    """the ranks' only side channel besides the communicator: files"""
    open(os.path.join(directory, f"at_{tag}_{rank}"), "w").close()
    t0 = time.time()
    while not all(os.path.exists(os.path.join(directory, f"at_{tag}_{r}")) for r in range(world)):
        if time.time() - t0 > 120:
            raise RuntimeError(f"barrier {tag}: a rank is missing")
        time.sleep(0.005)


def main():
    rank, world, directory = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = _lib.load()
    assert L.havac_gather_use_library(STANDIN.encode()) == 0
    assert L.havac_gather_use_library(STANDIN.encode()) == 0                    # the same path again: fine
    standin = C.CDLL(STANDIN)                                                   # the same handle the library has bound
    standin.standin_set_timeout_ms.argtypes = [C.c_uint32]
    standin.standin_set_delay_ms.argtypes = [C.c_uint32]
    out = {"rank": rank, "version": RcclGather.rccl_version()}
    stream = torch.cuda.current_stream(dev).cuda_stream
    g = RcclGather(rank, world, exchange_id(rank, directory, "a"))
    assert L.havac_gather_use_library(b"librccl.so.1") == _lib.E_LOGIC          # bound: no second library in this process

    # ---- 1. a real sharded pass: every rank sweeps its shard, rank 0 receives the whole list in device order ------------------
    nseg, nrows = 24, 300
    ncols = nseg * synth.SEGMENT
    model, cons = synth.dfam_like_model(nrows, 77)
    sym = synth.random_symbols(ncols, 78)
    synth.plant_homologs(sym, cons, ncols, every=7000, length=220)
    d_seq = torch.from_numpy(synth.pack_2bit(sym)).to(dev)
    d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
    hits = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    ctx = SsvContext()
    ctx.enqueue(d_seq.data_ptr(), ncols, d_phmm.data_ptr(), nrows, hits.data_ptr(), hits.numel(), rank, world, 0, stream)
    found = ctx.finish()
    counts = g.counts(found, stream)
    total = sum(counts)
    merged = torch.zeros(total + 16, dtype=torch.int64, device=dev) if rank == 0 else None
    g.records(hits.data_ptr(), merged.data_ptr() if rank == 0 else 0, merged.numel() if rank == 0 else 0, stream)
    g.wait()
    torch.cuda.synchronize(dev)
    lo, hi = shard_columns(ncols, rank, world)
    mine = hits[:found].cpu().numpy().view(np.uint64)
    cols = ((mine >> np.uint64(14)) & np.uint64(0x3FFFFFF)) * np.uint64(synth.SEGMENT) + (mine & np.uint64(0x3FFF))
    out["pass"] = {"found": int(found), "counts": counts, "inside_my_columns": bool(((cols >= lo) & (cols < hi)).all())}
    if rank == 0:
        want = O.ssv_fast(sym, model)
        got = merged[:total].cpu().numpy().view(np.uint64)
        out["pass"]["equals_oracle"] = bool(np.array_equal(got, want))
        out["pass"]["records"] = int(want.size)
        out["pass"]["tail_untouched"] = int(merged[total:].abs().sum().item()) == 0

    # ---- 2. synthetic lists: zero-length ranks, one rank with everything, a list of many chunks, rank 0 empty -------------------
    shapes = {"mixed": [3, 0, 100_000, 17, 0, 1, 70_000, 5], "all_in_last": [0] * 7 + [250_000], "rank0_empty": [0, 9, 0, 40_000, 1, 1, 0, 2],
              "nothing": [0] * 8, "big": [300_000, 1_200_000, 5, 650_000, 0, 90_000, 1, 1]}
    out["lists"] = {}
    for name, sizes in shapes.items():
        sizes = sizes[:world]
        if name == "all_in_last":
            sizes = [0] * (world - 1) + [250_000]
        mine = (np.arange(sizes[rank], dtype=np.uint64) * np.uint64(2654435761) + np.uint64(rank) * np.uint64(1 << 40)) | np.uint64(1)
        src = torch.from_numpy(mine.view(np.int64).copy()).to(dev)
        counts = g.counts(sizes[rank], stream)
        ok = counts == sizes
        dst = torch.full((sum(sizes) + 8,), -1, dtype=torch.int64, device=dev) if rank == 0 else None
        g.records(src.data_ptr(), dst.data_ptr() if rank == 0 else 0, dst.numel() if rank == 0 else 0, stream)
        g.wait()
        torch.cuda.synchronize(dev)
        if rank == 0:
            want = np.concatenate([(np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(r) * np.uint64(1 << 40)) | np.uint64(1)
                                   for r, n in enumerate(sizes)]) if sum(sizes) else np.zeros(0, np.uint64)
            got = dst.cpu().numpy()
            ok = ok and bool(np.array_equal(got[: sum(sizes)].view(np.uint64), want)) and bool((got[sum(sizes):] == -1).all())
        out["lists"][name] = bool(ok)

    # ---- 3. a rank whose pass failed (-1): every rank skips the records together, the communicator stays good ---------------------
    counts = g.counts(-1 if rank == world - 1 else 5, stream)
    try:
        g.records(hits.data_ptr(), hits.data_ptr(), hits.numel(), stream)
        out["failed_rank"] = "no error"
    except RuntimeError as e:
        out["failed_rank"] = {"counts": counts, "message": str(e)}
    counts = g.counts(2, stream)
    dst = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    g.records(hits.data_ptr(), dst.data_ptr(), dst.numel(), stream)
    g.wait()
    out["after_failed_rank"] = counts
    file_barrier(rank, world, directory, "three")

    # ---- 4. a receive buffer that is too small: refused on rank 0 before anything is posted; the others' sends never complete ----
    standin.standin_set_timeout_ms(1500)
    g.counts(100_000, stream)            # four chunks of the stand-in's mailbox: a send that nobody receives cannot complete
    t0 = time.time()
    try:
        g.records(hits.data_ptr(), hits.data_ptr(), 10, stream)      # (capacity only matters on rank 0)
        out["refused"] = "no error"
    except LengthError as e:
        out["refused"] = {"kind": "length", "message": str(e)}
    except RuntimeError as e:
        out["refused"] = {"kind": "runtime", "message": str(e), "seconds": round(time.time() - t0, 2)}
    try:
        g.counts(1, stream)
        out["after_refused"] = "usable"
    except Exception as e:  # noqa: BLE001
        out["after_refused"] = type(e).__name__
    g.close()            # aborts the broken communicator
    standin.standin_set_timeout_ms(60000)
    file_barrier(rank, world, directory, "four")

    # ---- 5. the deadline: an exchange that is still in flight when it passes -> HAVAC_E_TIMEOUT naming rank and stage -----------
    g = RcclGather(rank, world, exchange_id(rank, directory, "b"))
    g.set_deadline(5000)
    out["deadline_not_reached"] = g.counts(rank, stream)                          # an ordinary exchange under a deadline
    g.set_deadline(300)
    standin.standin_set_delay_ms(1500)                                            # a bounded kernel keeps the stream busy for 1.5 s
    t0 = time.time()
    try:
        g.counts(7, stream)
        out["deadline"] = "no error"
    except CollectiveTimeout as e:
        out["deadline"] = {"message": str(e), "seconds": round(time.time() - t0, 2)}
    torch.cuda.synchronize(dev)                                                   # the bounded kernel has ended by itself
    g.close()
    with open(os.path.join(directory, f"result_{rank}.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
