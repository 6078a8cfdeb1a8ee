"""The N>1 path on CPU: world_size-2 (and 3) gloo groups run havac_amd.dist.gather_hits on shards.

No GPU here, and the product has no CPU compute path, so each rank's shard of hits is produced by the
CPU checker (test infrastructure) filtered to the columns havac_ssv_shard_columns assigns to that
rank -- exactly the records the rank's GPU reports, in the same order (tests/test_gpu_scale.py proves
that on the GPU).  What is under test is everything after the kernel: shard arithmetic, the two
collectives, and that concatenation in rank order is already the reference's device order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker(rank, world, port, case, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from havac_amd import synth
    from havac_amd.dist import FAILED, ShardFailure, gather_hits
    from havac_amd.ssv import shard_columns
    from oracle import pyoracle as O

    if case == "empty":
        model = np.full((40, 4), -128, np.int8)
        sym = synth.random_symbols(synth.SEGMENT, 1)
    elif case == "lopsided":               # all hits in the columns of the last shard
        model, cons = synth.dfam_like_model(300, 3)
        sym = synth.random_symbols(6 * synth.SEGMENT, 4)
        sym[: 5 * synth.SEGMENT] = 0
        model[:, 0] = -100
        synth.plant_homologs(sym[5 * synth.SEGMENT:], cons, synth.SEGMENT, every=3000, length=280)
    elif case == "hundred_to_one":         # rank 0's columns are dense with hits, the last rank's almost empty
        model, cons = synth.dfam_like_model(120, 5)
        sym = synth.random_symbols(8 * synth.SEGMENT, 6)
        half = sym.size // 2
        synth.plant_homologs(sym[:half], cons, half, every=150, length=120, sub=0.02)
        synth.plant_homologs(sym[half:], cons, half, every=40000, length=16, sub=0.0, seed=9)
    else:
        model, cons = synth.model_collection([200, 900, 64], 11)
        sym = synth.random_symbols(12 * synth.SEGMENT, 12)
        synth.plant_homologs(sym, cons, sym.size, every=9000, length=500)
    whole = O.ssv(sym, model)
    rows, cols = O.unpack_hits(whole)
    lo, hi = shard_columns(sym.size, rank, world)
    mine = whole[(cols >= lo) & (cols < hi)]                      # a slice of the device-ordered whole
    cap = mine.size + 8                                         # a rank's buffer only has to hold its own records
    local = torch.zeros(cap, dtype=torch.int64)
    local[: mine.size] = torch.from_numpy(mine.view(np.int64))
    if case == "one_rank_fails":
        # a rank whose pass raised reports FAILED; every rank must come out of the collective and raise
        try:
            gather_hits(local, FAILED if rank == world - 1 else int(mine.size))
            raised = False
        except ShardFailure:
            raised = True
        if rank == 0:
            np.save(out_path, np.array([int(raised), 0, 0, 0]))
        assert raised
        dist.barrier()
        dist.destroy_process_group()
        return
    merged, counts = gather_hits(local, int(mine.size))
    assert counts[rank] == mine.size and sum(counts) == whole.size
    if rank == 0:
        got = merged.numpy().view(np.uint64)                     # no sort on rank 0: rank order is device order
        np.save(out_path, np.array([int(np.array_equal(got, whole)), whole.size, max(counts), min(counts)]))
    else:
        assert merged is None
    if rank == 0 and case == "mixed":
        # a receive buffer handed in is used as is (no second allocation, nothing concatenated)
        mine_again = torch.zeros(whole.size + 100, dtype=torch.int64)
    else:
        mine_again = None
    merged2, _ = gather_hits(local, int(mine.size), out=mine_again)
    if rank == 0 and mine_again is not None:
        assert merged2.data_ptr() == mine_again.data_ptr() and torch.equal(merged2, merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "mixed"), (3, "mixed"), (2, "empty"), (2, "lopsided"),
                                        (2, "hundred_to_one"), (3, "one_rank_fails")])
def test_gather_of_shards_over_gloo(tmp_path, oracle, world, case):
    out = str(tmp_path / "result.npy")
    mp.spawn(worker, args=(world, free_port(), case, out), nprocs=world, join=True)
    ok, total, most, least = np.load(out)
    assert ok == 1
    if case == "hundred_to_one":
        assert least > 0 and most >= 100 * least
    if case == "empty":
        assert total == 0
    if case == "mixed":
        assert total > 50 and least > 0
    if case == "lopsided":
        assert total > 0 and least == 0
