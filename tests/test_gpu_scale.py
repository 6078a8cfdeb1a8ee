"""Parity at BASELINE.json's shapes through size-independent properties: oracle windows with a left halo,
shard unions, repeatability.  The oracle finishes each window in seconds."""
import numpy as np
import pytest

from havac_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    return torch, torch.device("cuda", 0)


def run_shards(torch, dev, packed, model, world=1, capacity=1 << 22, tuning=None, orderings=None, variant=None, variants_run=None):
    """-> list of per-shard device-ordered record arrays (numpy uint64); `tuning` = keyword arguments of
    SsvContext.set_tuning; `orderings` (a list) receives last_ordering() of every shard's pass; `variant`: the kernel
    instantiation to force (SsvContext.set_kernel_variant), `variants_run` (a list) receives what every pass ran"""
    from havac_amd.ssv import SsvContext
    ctx = SsvContext()
    if tuning:
        ctx.set_tuning(**tuning)
    if variant is not None:
        ctx.set_kernel_variant(variant)
    d_seq = torch.from_numpy(packed).to(dev)
    d_phmm = torch.from_numpy(np.ascontiguousarray(model).reshape(-1)).to(dev)
    hits = torch.empty(capacity, dtype=torch.int64, device=dev)
    out = []
    stream = torch.cuda.current_stream(dev).cuda_stream
    for r in range(world):
        ctx.enqueue(d_seq.data_ptr(), packed.size * 4, d_phmm.data_ptr(), model.shape[0], hits.data_ptr(), capacity, r, world, 0, stream)
        n = ctx.finish()
        out.append(hits[:n].cpu().numpy().view(np.uint64).copy())
        if orderings is not None:
            orderings.append(ctx.last_ordering())
        if variants_run is not None:
            variants_run.append(ctx.last_kernel_variant())
    ctx.close()
    return out


def check_windows(oracle, packed, model, got, windows):
    rows, cols = oracle.unpack_hits(got)
    nrows = model.shape[0]
    for lo, hi in windows:
        start = max(0, lo - (nrows - 1)) // 4 * 4                         # whole packed bytes; a longer halo is harmless
        sym = synth.unpack_2bit(packed[start // 4: (hi + 3) // 4])[: hi - start]
        want = oracle.ssv_window(sym, model, lo - start, hi - start)
        wr, wc = oracle.unpack_hits(want)
        want = oracle.pack_hits(wr, wc + np.uint64(start))
        mine = got[(cols >= lo) & (cols < hi)]
        assert np.array_equal(oracle.device_order(mine), oracle.device_order(want)), (lo, hi, mine.size, want.size)


def whole_list(oracle, packed, model, lo=0, hi=None, cap=1 << 22):
    """Every record of columns [lo, hi) from the oracle's vectorised route (lo a multiple of 4)."""
    import os
    hi = packed.size * 4 if hi is None else hi
    start = max(0, lo - (model.shape[0] - 1)) // 4 * 4
    sym = synth.unpack_2bit(packed[start // 4: (hi + 3) // 4])[: hi - start]
    recs = oracle.ssv_fast(sym, model, nthreads=min(16, os.cpu_count() or 1), cap=cap)
    rows, cols = oracle.unpack_hits(recs)
    keep = cols + np.uint64(start) >= np.uint64(lo)
    return oracle.device_order(oracle.pack_hits(rows[keep], cols[keep] + np.uint64(start)))


def test_c2_full_size_windows_and_repeatability(torch_dev, oracle):
    """Config C2: L=1024 x 100,012,032 columns in one launch; exact check on windows spread over the matrix
    (including both ends and segment boundaries), hit count plausibility, and run-to-run identity."""
    torch, dev = torch_dev
    model, cons = synth.dfam_like_model(1024, synth.SEED_MODEL)
    packed = synth.random_packed(100_012_032, synth.SEED_SEQUENCE)
    got = run_shards(torch, dev, packed, model)[0]
    assert 500_000 < got.size < 2_000_000
    assert np.array_equal(got, oracle.device_order(got))                   # already in device order
    assert np.unique(got).size == got.size
    n = 100_012_032
    windows = [(0, 40_000), (n - 40_000, n), (12288 * 4000 - 20_000, 12288 * 4000 + 20_000)]
    rng = np.random.default_rng(5)
    windows += [(int(a) * 4, int(a) * 4 + 30_000) for a in rng.integers(1000, n // 4 - 10_000, size=5)]
    check_windows(oracle, packed, model, got, windows)
    again = run_shards(torch, dev, packed, model)[0]
    assert np.array_equal(got, again)


def test_c2_full_size_whole_hit_list(torch_dev, oracle):
    """Config C2, every one of its 1.02e11 cells: the complete hit list of the launch against the oracle's
    vectorised whole-matrix route (oracle.ssv_fast; itself pinned to the plain sweep and the reference object in
    tests/test_oracle.py).  Element for element, in device order."""
    import os
    torch, dev = torch_dev
    model, cons = synth.dfam_like_model(1024, synth.SEED_MODEL)
    packed = synth.random_packed(100_012_032, synth.SEED_SEQUENCE)
    got = run_shards(torch, dev, packed, model)[0]
    sym = synth.unpack_2bit(packed)
    want = oracle.ssv_fast(sym, model, nthreads=min(16, os.cpu_count() or 1), cap=1 << 21)
    assert got.size == want.size
    assert np.array_equal(got, want)


def test_c3_shape_whole_hit_list(torch_dev, oracle):
    """Config C3's shape (hundreds of concatenated models) at a size the vectorised oracle finishes in seconds:
    ~60k rows x 1.2 Mbp = 7.4e10 cells, whole list."""
    import os
    torch, dev = torch_dev
    model, cons = synth.model_collection(synth.model_lengths(120), 2101)
    n = 100 * synth.SEGMENT
    packed = synth.random_packed(n, 1303)
    got = run_shards(torch, dev, packed, model, capacity=1 << 23)[0]
    want = oracle.ssv_fast(synth.unpack_2bit(packed), model, nthreads=min(16, os.cpu_count() or 1), cap=1 << 22)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shard_union_equals_whole(torch_dev, oracle, world):
    """The column shards the multi-GPU path uses: concatenated in shard order they ARE the single-launch answer."""
    torch, dev = torch_dev
    from havac_amd.ssv import shard_columns
    model, cons = synth.model_collection([300, 50, 1200, 700], 31)
    sym = synth.random_symbols(30 * synth.SEGMENT, 77)
    synth.plant_homologs(sym, cons, sym.size, every=20_000, length=600)
    packed = synth.pack_2bit(sym)
    whole = run_shards(torch, dev, packed, model)[0]
    want = oracle.ssv_mt(sym, model)
    assert np.array_equal(whole, want)
    parts = run_shards(torch, dev, packed, model, world=world)
    for r, part in enumerate(parts):
        rows, cols = oracle.unpack_hits(part)
        lo, hi = shard_columns(sym.size, r, world)
        assert ((cols >= lo) & (cols < hi)).all()
    assert np.array_equal(np.concatenate(parts), whole)           # no sort needed: shard order is device order


def test_c5_long_model_windows(torch_dev, oracle):
    """Config C5 shape: one model of 20000 rows (far more than one wave tile is wide) x 10 Mbp; windows with
    the full 19999-column halo."""
    torch, dev = torch_dev
    model, cons = synth.dfam_like_model(20000, 2005)
    n = 814 * synth.SEGMENT
    packed = synth.random_packed(n, 1005)
    got = run_shards(torch, dev, packed, model, capacity=1 << 23)[0]
    assert got.size > 100_000
    check_windows(oracle, packed, model, got, [(0, 3000), (n - 3000, n), (5_000_000, 5_003_000)])
    assert np.array_equal(got, whole_list(oracle, packed, model))         # all 2.0e11 cells


def test_c3_full_size_whole_hit_list(torch_dev, oracle):
    """Config C3 as BASELINE.json states it: 1000 models (lengths log-uniform in 50..2000, ~5e5 rows in all) x 10 Mbp,
    one launch, ~5e12 cells: every record against the oracle's vectorised route."""
    torch, dev = torch_dev
    model, cons = synth.model_collection(synth.model_lengths(1000), 2101)
    n = 814 * synth.SEGMENT
    packed = synth.random_packed(n, 1303)
    got = run_shards(torch, dev, packed, model, capacity=1 << 27)[0]
    assert got.size > 10_000_000
    want = whole_list(oracle, packed, model, cap=got.size + 1024)
    assert got.size == want.size and np.array_equal(got, want)


def test_c5_full_size_whole_hit_list(torch_dev, oracle):
    """Config C5 as BASELINE.json states it: one model of 20000 rows x 100 Mbp (2.0e12 cells), every record."""
    torch, dev = torch_dev
    model, cons = synth.dfam_like_model(20000, 2005)
    packed = synth.random_packed(100_012_032, 1005)
    got = run_shards(torch, dev, packed, model, capacity=1 << 26)[0]
    assert got.size > 1_000_000
    want = whole_list(oracle, packed, model, cap=got.size + 1024)
    assert got.size == want.size and np.array_equal(got, want)


def test_sort_hits_entry_point(torch_dev, oracle):
    """havac_ssv_sort_hits: any list of packed records -> the reference's device order, in place."""
    torch, dev = torch_dev
    from havac_amd.ssv import SsvContext
    rng = np.random.default_rng(12)
    n = 300_000
    rows = rng.integers(0, 1 << 24, size=n, dtype=np.uint64)
    cols = rng.integers(0, 1 << 32, size=n, dtype=np.uint64)
    recs = oracle.pack_hits(rows, cols)
    t = torch.from_numpy(recs.view(np.int64).copy()).to(dev)
    ctx = SsvContext()
    ctx.sort_hits(t.data_ptr(), n, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    got = t.cpu().numpy().view(np.uint64)
    ctx.close()
    assert np.array_equal(got, oracle.device_order(recs))


def test_ordering_more_than_2_to_32_records():
    """C4 on ONE GPU orders 4.5e9 records in one list.  4.4e9 random records through havac_ssv_sort_hits: device order,
    same multiset.  (Round 2 found the key <-> record kernels launched with one thread per record: a dispatch counts its
    work-items in 32 bits, so beyond 2^32 records the tail of the list stayed unconverted.  They are grid-stride now.)
    HIP through ctypes and numpy on the host: torch's own kernels mis-index tensors of more than 2^32 elements here."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "big_sort_check.py"), "4.4e9"], capture_output=True,
                       text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "4400000000 records: same multiset True, out-of-order neighbours 0" in r.stdout, r.stdout[-500:]


def test_columns_beyond_2_to_32(torch_dev, oracle):
    """5.0e9 columns (1.25 GB packed): more than the reference's 32-bit symbol count allows
    (host/HavacHwClient.cpp:81), inside the record format's 26-bit segment field.  Windows at both ends."""
    torch, dev = torch_dev
    nseg = 407_000                                              # 5,001,216,000 columns
    n = nseg * synth.SEGMENT
    assert n > (1 << 32)
    packed = synth.random_packed(n, 4242)
    model, cons = synth.dfam_like_model(96, 4243)
    got = run_shards(torch, dev, packed, model, capacity=1 << 23)[0]
    _, cols = oracle.unpack_hits(got)
    assert got.size > 100_000 and int(cols.max()) > (1 << 32)
    assert np.array_equal(got, oracle.device_order(got))
    check_windows(oracle, packed, model, got, [(0, 200_000), (n - 200_000, n), ((1 << 32) - 100_000, (1 << 32) + 100_000)])


def test_c4_shape_one_shard_of_eight(torch_dev, oracle):
    """Config C4's sharding at reduced height: a 1 Gbp database, 3000 model rows, shard 5 of 8 (what rank 5 of an
    8-GPU node computes); windows inside the shard, at its two edges (halo side and cut side), none outside."""
    torch, dev = torch_dev
    from havac_amd.ssv import SsvContext, shard_columns
    n = 81_381 * synth.SEGMENT                                  # 1,000,009,728 columns
    packed = synth.random_packed(n, 777)
    model, cons = synth.model_collection([900, 1300, 800], 778)
    lo, hi = shard_columns(n, 5, 8)
    ctx = SsvContext()
    d_seq = torch.from_numpy(packed).to(dev)
    d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
    cap = 1 << 24
    hits = torch.empty(cap, dtype=torch.int64, device=dev)
    ctx.enqueue(d_seq.data_ptr(), n, d_phmm.data_ptr(), model.shape[0], hits.data_ptr(), cap, 5, 8, 0,
                torch.cuda.current_stream(dev).cuda_stream)
    found = ctx.finish()
    got = hits[:found].cpu().numpy().view(np.uint64).copy()
    ctx.close()
    _, cols = oracle.unpack_hits(got)
    assert found > 1_000_000 and int(cols.min()) >= lo and int(cols.max()) < hi
    assert np.array_equal(got, oracle.device_order(got))
    check_windows(oracle, packed, model, got, [(lo, lo + 20_000), (hi - 20_000, hi), ((lo + hi) // 2 // 4 * 4, (lo + hi) // 2 // 4 * 4 + 20_000)])
    assert np.array_equal(got, whole_list(oracle, packed, model, lo, hi))  # the shard's 3.75e11 cells, every record


def test_c4_full_height_one_shard_of_eight(torch_dev, oracle):
    """Config C4 as BASELINE.json states it, as far as one GPU goes: the 1 Gbp database x the 1000-model collection
    (503,329 rows), the columns rank 5 of an 8-GPU node owns -- 1.25e8 columns, 6.3e13 cells, 5.6e8 records -- and
    every record of two 1.5e7-column stretches of it (the shard's left edge, where the halo ends, and its middle;
    6.7e7 records each) compared with the checker.  The other seven ranks run the same code on other columns."""
    torch, dev = torch_dev
    from havac_amd.ssv import SsvContext, shard_columns
    n = 81_381 * synth.SEGMENT                                  # 1,000,009,728 columns
    model, cons = synth.model_collection(synth.model_lengths(1000), 2101)
    nrows = model.shape[0]
    assert nrows > 500_000
    packed = synth.random_packed(n, 777)
    lo, hi = shard_columns(n, 5, 8)
    d_seq = torch.from_numpy(packed).to(dev)
    d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
    cap = 1 << 30
    hits = torch.empty(cap, dtype=torch.int64, device=dev)
    ctx = SsvContext()
    ctx.enqueue(d_seq.data_ptr(), n, d_phmm.data_ptr(), nrows, hits.data_ptr(), cap, 5, 8, 0,
                torch.cuda.current_stream(dev).cuda_stream)
    found = ctx.finish()
    ctx.close()
    assert found > 500_000_000
    rec = hits[:found]
    cols = ((rec >> 14) & 0x3FFFFFF) * synth.SEGMENT + (rec & 0x3FFF)
    assert int(cols.min()) >= lo and int(cols.max()) < hi
    assert bool((rec[1:] != rec[:-1]).all())                                  # no record twice
    for a in (lo, (lo + hi) // 2 // 4 * 4):
        b = a + 15_000_000
        mine = rec[(cols >= a) & (cols < b)].cpu().numpy().view(np.uint64)
        assert np.array_equal(mine, oracle.device_order(mine))             # a slice of the device-ordered list
        start = max(0, a - (nrows - 1)) // 4 * 4
        sym = synth.unpack_2bit(packed[start // 4: b // 4])
        want = oracle.ssv_fast(sym, model, nthreads=16, cap=mine.size + (1 << 24))
        rows_w, cols_w = oracle.unpack_hits(want)
        keep = cols_w + np.uint64(start) >= np.uint64(a)
        want = oracle.device_order(oracle.pack_hits(rows_w[keep], cols_w[keep] + np.uint64(start)))
        assert mine.size > 60_000_000 and np.array_equal(mine, want)
    del hits, rec, cols
    torch.cuda.empty_cache()


def test_passes_in_flight_give_the_same_lists(torch_dev, oracle):
    """ShardedSsv with several passes in flight (bench.py's default): six different problems submitted through
    three slots come back in submission order, each equal to its own oracle list."""
    torch, dev = torch_dev
    from havac_amd.dist import ShardedSsv
    eng = ShardedSsv(1 << 20, dev, depth=3)
    problems = []
    for k in range(6):
        model, cons = synth.model_collection([200 + 37 * k, 90], 600 + k)
        sym = synth.random_symbols((3 + k) * synth.SEGMENT, 700 + k)
        synth.plant_homologs(sym, cons, sym.size, every=9000, length=150, seed=k)
        problems.append((torch.from_numpy(synth.pack_2bit(sym)).to(dev), sym.size,
                         torch.from_numpy(model.reshape(-1)).to(dev), model.shape[0], oracle.ssv_mt(sym, model)))
    got = []
    for d_seq, n, d_phmm, nrows, _ in problems:
        if len(eng.in_flight) == 3:
            hits, found = eng.collect()
            got.append(hits.cpu().numpy().view(np.uint64).copy())
        eng.submit(d_seq, n, d_phmm, nrows)
    while eng.in_flight:
        hits, found = eng.collect()
        got.append(hits.cpu().numpy().view(np.uint64).copy())
    with pytest.raises(RuntimeError):
        for _ in range(4):
            eng.submit(*problems[0][:4])
    while eng.in_flight:
        eng.collect()
    eng.close()
    assert len(got) == 6
    for (_, _, _, _, want), mine in zip(problems, got):
        assert want.size > 0 and np.array_equal(mine, want)


@pytest.mark.parametrize("nrows", [32, 64, 100, 256])
def test_short_models_whole_hit_list(torch_dev, oracle, nrows):
    """Short single models x 100 Mbp: the resident-table kernel (1,536 workgroups, every wave walks 7 or 8 adjacent tiles, the
    model's tables built once per workgroup) and the standard kernel (tiles of one to eight chunks, 12,209 blocks that end within
    microseconds of each other and leave their records through the block tails); the whole hit list against the checker."""
    torch, dev = torch_dev
    ncols = 100_012_032
    model, cons = synth.dfam_like_model(nrows, 4242 + nrows)
    packed = synth.random_packed(ncols, 4243)
    ran = []
    got, = run_shards(torch, dev, packed, model, variants_run=ran)
    assert ran == [1]                    # the resident-table kernel (round 4)
    assert got.size > 1000
    want = whole_list(oracle, packed, model)
    assert np.array_equal(got, want)
    # the standard kernel on the same problem (single tiles), and the standard kernel walking groups of four
    again, = run_shards(torch, dev, packed, model, variant=0, variants_run=ran)
    assert ran[-1] == 0 and np.array_equal(again, want)
    if nrows == 64:
        again, = run_shards(torch, dev, packed, model, tuning=dict(tiles_per_item=-4), variant=0)
        assert np.array_equal(again, want)
        # the resident-table kernel with other runs: every tile a wave of its own, runs of 2 and of 50 tiles
        for walk in (1, 2, 50):
            again, = run_shards(torch, dev, packed, model, tuning=dict(tiles_per_item=walk), variant=1, variants_run=ran)
            assert ran[-1] == 1 and np.array_equal(again, want), walk


def test_short_model_kernel_on_small_ragged_problems(torch_dev, oracle):
    """ssv_resident_kernel where its runs and its prefetch meet the matrix's edges: a handful of tiles (the first tile's diagonals
    start left of column 0, the last one's end right of column N), one-row and 256-row models (one to nine tables), runs of 1 to
    8 tiles and tile counts that are no multiple of the run, dense-hit models (two-step windows), sharded runs (every shard
    walks its own tiles), and the cases it must NOT take: 257 rows, a separator mask (boundary mode: test_gpu_boundary_mode.py)
    -- the library's choice falls back to the standard kernel."""
    torch, dev = torch_dev
    rng = np.random.default_rng(404)
    for case in range(24):
        nrows = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 96, 100, 127, 128, 129, 200, 255, 256]))
        nseg = int(rng.integers(1, 9))
        sym = synth.random_symbols(nseg * synth.SEGMENT, 150 + case)
        if case % 4 == 0:
            model = rng.integers(-128, 128, size=(nrows, 4)).astype(np.int8)
        else:
            model, cons = synth.dfam_like_model(nrows, 160 + case)
            synth.plant_homologs(sym, cons, sym.size, every=700, length=min(nrows, 150), sub=0.08)
        walk = int(rng.choice([-1, 1, 2, 3, 4, 8, 5, 7]))
        ran = []
        got, = run_shards(torch, dev, synth.pack_2bit(sym), model, capacity=1 << 23, tuning=dict(tiles_per_item=walk), variant=1, variants_run=ran)
        want = oracle.ssv(sym, model, cap=1 << 23)
        assert ran == [1] and np.array_equal(got, want), (case, nrows, nseg, walk, got.size, want.size)
    sym = synth.random_symbols(11 * synth.SEGMENT, 199)
    model, cons = synth.dfam_like_model(96, 198)
    synth.plant_homologs(sym, cons, sym.size, every=700, length=60, sub=0.05)
    ran = []
    parts = run_shards(torch, dev, synth.pack_2bit(sym), model, world=4, variants_run=ran)
    assert ran == [1, 1, 1, 1] and np.array_equal(np.concatenate(parts), oracle.ssv(sym, model))
    model257, _ = synth.dfam_like_model(257, 197)
    ran = []
    got, = run_shards(torch, dev, synth.pack_2bit(sym), model257, variant=1, variants_run=ran)
    assert ran == [0] and np.array_equal(got, oracle.ssv(sym, model257))


@pytest.mark.parametrize("per_item", [2, 3, 8])
def test_tiles_per_item_forced_on_small_problems(torch_dev, oracle, per_item):
    """the same path on small, ragged problems (tile counts that are no multiple of the group, one-row models, the
    matrix's edges inside a group): havac_ssv_set_tuning forces the grouping (off by default, kept for experiments)"""
    torch, dev = torch_dev
    tuning = dict(tiles_per_item=per_item)
    rng = np.random.default_rng(per_item)
    for case in range(12):
        nrows = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 128, 300]))
        nseg = int(rng.integers(1, 8))
        sym = synth.random_symbols(nseg * synth.SEGMENT, 50 + case)
        if case % 3 == 0:
            model = rng.integers(-128, 128, size=(nrows, 4)).astype(np.int8)       # dense hits, two-step windows
        else:
            model, cons = synth.dfam_like_model(nrows, 60 + case)
            synth.plant_homologs(sym, cons, sym.size, every=900, length=min(nrows, 150), sub=0.08)
        got, = run_shards(torch, dev, synth.pack_2bit(sym), model, capacity=1 << 23, tuning=tuning)
        want = oracle.ssv(sym, model, cap=1 << 23)
        assert np.array_equal(got, want), (per_item, case, nrows, nseg, got.size, want.size)
    # and sharded: every shard groups its own tiles
    sym = synth.random_symbols(9 * synth.SEGMENT, 99)
    model, cons = synth.dfam_like_model(64, 98)
    synth.plant_homologs(sym, cons, sym.size, every=700, length=60, sub=0.05)
    parts = run_shards(torch, dev, synth.pack_2bit(sym), model, world=4, tuning=tuning)
    assert np.array_equal(np.concatenate(parts), oracle.ssv(sym, model))


@pytest.mark.parametrize("tails", [0, 1, 2])
def test_block_tails_on_and_off(torch_dev, oracle, tails):
    """What a block still has staged at its end leaves through a side buffer and a gather kernel (block tails: the default for
    items of up to 512 rows, havac_ssv_set_tuning(block_tails=2) forces them for taller ones) or with one returning atomic per block
    (block_tails=0, tall items, launches too big for a side buffer): the same lists either way, on sparse hits (every tail fits), dense hits (tails that overflow their 128 slots
    fall back to the atomic) and a short model x 100 Mbp."""
    torch, dev = torch_dev
    tuning = dict(block_tails=tails)
    rng = np.random.default_rng(17)
    for case in range(8):
        nrows = int(rng.choice([1, 32, 64, 100, 1000, 3000]))
        nseg = int(rng.integers(1, 12))
        sym = synth.random_symbols(nseg * synth.SEGMENT, 500 + case)
        if case % 2 == 0:
            model = rng.integers(-100, 128, size=(nrows, 4)).astype(np.int8)        # dense: hundreds of records per block
        else:
            model, cons = synth.dfam_like_model(nrows, 600 + case)
            synth.plant_homologs(sym, cons, sym.size, every=800, length=min(nrows, 200), sub=0.08)
        got, = run_shards(torch, dev, synth.pack_2bit(sym), model, capacity=1 << 24, tuning=tuning)
        want = oracle.ssv(sym, model, cap=1 << 24)
        assert np.array_equal(got, want), (tails, case, nrows, nseg, got.size, want.size)
    ncols = 100_012_032
    model, cons = synth.dfam_like_model(48, 4300)
    packed = synth.random_packed(ncols, 4301)
    got, = run_shards(torch, dev, packed, model, tuning=tuning)
    assert np.array_equal(got, whole_list(oracle, packed, model))


def test_ordering_by_buckets_equals_radix_sort_and_the_checker(torch_dev, oracle):
    """hit_order.hip.h: the records are put in the FPGA's emission order (device/HavacHls.cpp:151-152,264;
    device/HitReporting.cpp:178-337) by bucketing on (segment, row range) and sorting every bucket in LDS.  Whatever the
    shape -- one bucket per segment (short models), row ranges (tall ones), small and large buckets, shards, a single
    record, none -- the list equals the checker's element for element, and the radix sort's (ordering=0)."""
    torch, dev = torch_dev
    rng = np.random.default_rng(31)
    seen_paths, largest_seen = set(), 0
    for case in range(14):
        nrows = int(rng.choice([1, 40, 300, 1024, 5000, 20000]))
        nseg = int(rng.integers(1, 30))
        sym = synth.random_symbols(nseg * synth.SEGMENT, 900 + case)
        kind = case % 4
        if kind == 0:
            model = rng.integers(-110, 128, size=(nrows, 4)).astype(np.int8)                # dense: large buckets
        elif kind == 1:
            model = np.full((nrows, 4), -128, np.int8)                                      # nothing hits
            if case == 1:
                model[: min(nrows, 3)] = 127                                                # ... or every cell of row 2 does
        else:
            model, cons = synth.dfam_like_model(nrows, 700 + case)
            synth.plant_homologs(sym, cons, sym.size, every=1500, length=min(nrows, 250), sub=0.08)
        packed = synth.pack_2bit(sym)
        want = oracle.ssv_mt(sym, model, cap=1 << 25)
        for world in (1, 3):
            if world > nseg:
                continue
            orderings = []
            by_buckets = run_shards(torch, dev, packed, model, world=world, capacity=1 << 25, orderings=orderings)
            by_radix = run_shards(torch, dev, packed, model, world=world, capacity=1 << 25, tuning=dict(ordering=0))
            assert np.array_equal(np.concatenate(by_buckets), want), (case, world, nrows, nseg, want.size)
            assert all(np.array_equal(a, b) for a, b in zip(by_buckets, by_radix))
            seen_paths.update(path for path, _, _ in orderings)
            largest_seen = max([largest_seen] + [big for _, _, big in orderings])
    assert 1 in seen_paths                      # the bucket ordering really ran
    assert largest_seen > 2048                  # ... through its large-bucket sorter too


def test_ordering_gives_up_on_a_bucket_too_big_for_lds(torch_dev, oracle):
    """One model of a sparse collection that hits everywhere: its buckets hold far more than the launch's average and
    more than an LDS sort takes (16,384 records); the pass then falls back to the radix sort -- same list."""
    torch, dev = torch_dev
    quiet = np.full((6000, 4), -128, np.int8)
    hot = np.full((40, 4), 127, np.int8)                          # every cell of 38 rows x all columns hits every third row
    model = np.concatenate([quiet[:3000], hot, quiet[3000:]])
    sym = synth.random_symbols(40 * synth.SEGMENT, 77)
    orderings = []
    got, = run_shards(torch, dev, synth.pack_2bit(sym), model, capacity=1 << 25, orderings=orderings)
    want = oracle.ssv_mt(sym, model, cap=1 << 25)
    assert want.size > 1_000_000 and np.array_equal(got, want)
    assert orderings[0][0] == 2 and orderings[0][2] > 16384      # given up for the radix sort; the largest bucket said why


def test_a_rank_needs_only_its_window_of_the_database(torch_dev, oracle):
    """havac_ssv_shard_window / havac_ssv_set_sequence_window: a rank of a sharded run holds the columns its shard READS
    (its own, the left halo, a few thousand for the tiling) and nothing else -- C4: 31 MB per rank instead of 250 MB.
    Every shard computed from exactly its window (the rest of the database is not on the device at all) gives the list
    the whole buffer gives; a window that is too small is refused, not read past."""
    from havac_amd.ssv import SsvContext, shard_columns, shard_window
    torch, dev = torch_dev
    nseg, world = 600, 5
    ncols = nseg * synth.SEGMENT
    packed = synth.random_packed(ncols, 314)
    model, cons = synth.model_collection([700, 90, 2000, 1300], 315)       # 4090 rows: halo of a third of a segment
    sym = synth.unpack_2bit(packed)
    synth.plant_homologs(sym, cons, sym.size, every=40_000, length=280, sub=0.08)
    packed = synth.pack_2bit(sym)
    want = oracle.ssv_fast(sym, model, nthreads=8, cap=1 << 23)
    d_phmm = torch.from_numpy(model.reshape(-1)).to(dev)
    hits = torch.empty(1 << 22, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ctx = SsvContext()
    parts, held = [], 0
    for r in range(world):
        first, end = shard_window(ncols, model.shape[0], r, world)
        lo, hi = shard_columns(ncols, r, world)
        assert first % synth.SEGMENT == 0 and end % synth.SEGMENT == 0 and first <= lo and hi <= end <= ncols
        assert lo - first <= model.shape[0] + 3 * synth.SEGMENT and end - hi <= 2 * synth.SEGMENT      # no more than it needs
        window = torch.from_numpy(packed[first // 4: end // 4].copy()).to(dev)       # only this much of the database exists on the GPU
        held += window.numel()
        ctx.set_sequence_window(first, end - first)
        ctx.enqueue(window.data_ptr(), ncols, d_phmm.data_ptr(), model.shape[0], hits.data_ptr(), hits.numel(), r, world, 0, stream)
        n = ctx.finish()
        parts.append(hits[:n].cpu().numpy().view(np.uint64).copy())
        del window
    assert np.array_equal(np.concatenate(parts), want)
    assert held < 1.3 * packed.size                         # five windows together: little more than one copy
    # too small a window: one segment short on the left
    first, end = shard_window(ncols, model.shape[0], 3, world)
    window = torch.from_numpy(packed[(first + synth.SEGMENT) // 4: end // 4].copy()).to(dev)
    ctx.set_sequence_window(first + synth.SEGMENT, end - first - synth.SEGMENT)
    with pytest.raises(ValueError, match="does not cover the columns this shard reads"):
        ctx.enqueue(window.data_ptr(), ncols, d_phmm.data_ptr(), model.shape[0], hits.data_ptr(), hits.numel(), 3, world, 0, stream)
    ctx.set_sequence_window(0, 0)
    ctx.close()


@pytest.mark.parametrize("split", [
    dict(),                                                       # the library's own rule
    dict(parts_log2=3, split_rounds_x4=1),                        # eight partitions, each with whole tiles first and cut tiles last
    dict(parts_log2=0, split_rounds_x4=1, short_rows=1024),       # one partition, finer cuts
    dict(parts_log2=3, split_rounds_x4=0),                        # partitions of whole tiles only
    dict(parts_log2=2, split_rounds_x4=64, short_rows=2048, guide=3),   # every tile cut
])
def test_partitions_whole_tiles_first_cut_tiles_last(torch_dev, oracle, split):
    """How a launch hands out its tiles (ssv_kernels.hip.h, "items"; havac_dev.hip, plan_row_cuts): partitions of adjacent
    tiles, whole tiles first, the last tiles cut by rows with cuts that get finer towards the model's end.  Whatever the
    split, the whole hit list equals the checker's."""
    torch, dev = torch_dev
    O = oracle
    from havac_amd.ssv import SsvContext
    model, cons = synth.dfam_like_model(8200, 77)
    ncols = 330 * synth.SEGMENT                                   # 1,980 tiles + the model's: more than 8 x 192 cut tiles
    sym = synth.random_symbols(ncols, 78)
    synth.plant_homologs(sym, cons, ncols, every=150_000, length=300)
    packed = synth.pack_2bit(sym)
    want = whole_list(O, packed, model, cap=1 << 24)
    ctx = SsvContext()
    ctx.set_split_tuning(**split)
    d_seq = torch.from_numpy(packed).to(dev)
    d_phmm = torch.from_numpy(np.ascontiguousarray(model).reshape(-1)).to(dev)
    capacity = 1 << 24
    hits = torch.empty(capacity, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(2):                                            # twice: tickets, flags and hand-off buffers are reused
        ctx.enqueue(d_seq.data_ptr(), packed.size * 4, d_phmm.data_ptr(), model.shape[0], hits.data_ptr(), capacity, 0, 1, 0, stream)
        n = ctx.finish()
        got = hits[:n].cpu().numpy().view(np.uint64)
        assert got.size == want.size and np.array_equal(got, want)
    ctx.close()


def test_row_cut_table_overflows_into_uniform_blocks(torch_dev, oracle):
    """A slow taper over a tall model needs more row blocks than the table of individual cuts holds (32): the rest are uniform
    blocks of the finest height.  65,536 rows, 1/16 of the remaining rows per block, finest block 1024 rows: 37 blocks."""
    torch, dev = torch_dev
    O = oracle
    from havac_amd.ssv import SsvContext
    model, cons = synth.dfam_like_model(65_500, 91)
    ncols = 10 * synth.SEGMENT
    sym = synth.random_symbols(ncols, 92)
    synth.plant_homologs(sym, cons, ncols, every=30_000, length=400)
    packed = synth.pack_2bit(sym)
    want = whole_list(O, packed, model, cap=1 << 24)
    ctx = SsvContext()
    ctx.set_split_tuning(parts_log2=3, split_rounds_x4=64, short_rows=1024, guide=16)
    d_seq = torch.from_numpy(packed).to(dev)
    d_phmm = torch.from_numpy(np.ascontiguousarray(model).reshape(-1)).to(dev)
    capacity = 1 << 24
    hits = torch.empty(capacity, dtype=torch.int64, device=dev)
    ctx.enqueue(d_seq.data_ptr(), packed.size * 4, d_phmm.data_ptr(), model.shape[0], hits.data_ptr(), capacity, 0, 1, 0,
                torch.cuda.current_stream(dev).cuda_stream)
    n = ctx.finish()
    got = hits[:n].cpu().numpy().view(np.uint64)
    assert got.size == want.size and np.array_equal(got, want)
    ctx.close()


def _synthetic_gathered_list(torch, dev, nseg, per_segment, world):
    """A device-ordered list of nseg * per_segment records, laid out like the gathered list of `world` ranks over a database
    of nseg segments: record i lies in segment i // per_segment; inside a segment rows ascend and, inside a row, columns.
    Filled in slices of 2^27 records (every torch kernel stays far below 2^32 elements) -> (tensor, counts, spans)."""
    from havac_amd.ssv import shard_columns
    n = nseg * per_segment
    t = torch.empty(n, dtype=torch.int64, device=dev)
    step = 1 << 27
    for a in range(0, n, step):
        b = min(n, a + step)
        i = torch.arange(a, b, dtype=torch.int64, device=dev)
        seg, j = i // per_segment, i % per_segment
        row, col = j >> 2, (j & 3) * 3000 + (j >> 2) % 1000
        t[a:b] = col | (seg << 14) | (row << 40)
    spans = [shard_columns(nseg * synth.SEGMENT, r, world) for r in range(world)]
    counts = [(hi - lo) // synth.SEGMENT * per_segment for lo, hi in spans]
    assert sum(counts) == n
    return t, counts, spans


def _swap(torch, t, i):
    pair = t[i: i + 2].clone()
    t[i: i + 2] = pair.flip(0)


@pytest.mark.parametrize("nseg,per_segment", [(977, 211), (81_381, 54_067)])
def test_gathered_list_check_survives_c4s_own_list(torch_dev, nseg, per_segment):
    """What rank 0 of `bench.py --gpus 8 --workload c4` asserts about the 4.46e9 records it has gathered (VERDICT round 3,
    item 1): through the C ABI (havac_ssv_check_order: one grid-stride HIP pass, 64-bit indices, no list-sized temporary) and
    through bench.py's own routine.  A synthetic device-ordered list of 4.40e9 records (35 GB) with eight rank spans on ONE
    GPU -> all flags true; one pair of records swapped beyond element 2^32 -> exactly the order flag false, at that index; the
    boundary between two ranks' stretches shifted by one record (a record of rank 5 counted as rank 4's) -> exactly the
    columns flag false, at that index.  The small case is the same with everything below 2^18."""
    torch, dev = torch_dev
    import importlib.util
    import os
    from havac_amd.ssv import check_order
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("havac_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    world = 8
    t, counts, spans = _synthetic_gathered_list(torch, dev, nseg, per_segment, world)
    n = int(t.numel())
    big = n > (1 << 32)
    assert big == (nseg > 1000) and (not big or n >= 4_400_000_000)
    torch.cuda.synchronize(dev)
    free_before, _ = torch.cuda.mem_get_info(dev)

    clean = check_order(t.data_ptr(), n, counts, spans)
    assert clean == {"records": n, "out_of_order": 0, "first_out_of_order": None, "out_of_span": 0, "first_out_of_span": None}
    out = bench.gathered_list_checks(t, counts, spans)
    assert out["counts_add_up"] and out["device_order_no_duplicates"] and out["ranks_inside_their_columns"]
    free_after, _ = torch.cuda.mem_get_info(dev)
    assert free_before - free_after < (64 << 20)            # no list-sized temporary was allocated (and kept) by the check

    # (a) two neighbours swapped -- beyond element 2^32 in the big case -- inside one segment
    i0 = ((1 << 32) + 123_456 if big else n // 2) // per_segment * per_segment + 7
    _swap(torch, t, i0)
    rep = check_order(t.data_ptr(), n, counts, spans)
    assert rep["out_of_order"] == 1 and rep["first_out_of_order"] == i0 + 1 and rep["out_of_span"] == 0
    out = bench.gathered_list_checks(t, counts, spans)
    assert out["device_order_no_duplicates"] is False and out["ranks_inside_their_columns"] is True
    assert out["out_of_order"] == {"records": 1, "first_index": i0 + 1}
    # a duplicate is out of order too
    saved = t[i0 + 5].clone()
    t[i0 + 5] = t[i0 + 4]
    assert check_order(t.data_ptr(), n)["out_of_order"] == 2
    t[i0 + 5] = saved
    _swap(torch, t, i0)
    assert check_order(t.data_ptr(), n, counts, spans)["out_of_order"] == 0

    # (b) the first record of rank 5 counted as rank 4's: it lies outside rank 4's columns
    moved = list(counts)
    moved[4] += 1
    moved[5] -= 1
    boundary = sum(counts[:5])
    assert not big or boundary > (1 << 31)
    rep = check_order(t.data_ptr(), n, moved, spans)
    assert rep["out_of_span"] == 1 and rep["first_out_of_span"] == boundary and rep["out_of_order"] == 0
    out = bench.gathered_list_checks(t, moved, spans)
    assert out["device_order_no_duplicates"] is True and out["ranks_inside_their_columns"] is False
    assert out["outside_their_rank"] == {"records": 1, "first_index": boundary}

    # (c) both at once; counts that do not add up are refused, not checked
    _swap(torch, t, i0)
    out = bench.gathered_list_checks(t, moved, spans)
    assert out["device_order_no_duplicates"] is False and out["ranks_inside_their_columns"] is False
    short = list(counts)
    short[7] -= 1
    assert bench.gathered_list_checks(t, short, spans)["counts_add_up"] is False
    with pytest.raises(Exception):
        check_order(t.data_ptr(), n, short, spans)
    del t
    torch.cuda.empty_cache()


def test_gathered_list_check_against_numpy(torch_dev, oracle):
    """havac_ssv_check_order on random lists with random damage, against the same definition in numpy: records out of the
    reference's emission order (device/HavacHls.cpp:151-152,264), duplicates, records outside their rank's columns."""
    torch, dev = torch_dev
    from havac_amd.ssv import check_order, shard_columns
    rng = np.random.default_rng(99)
    for trial in range(12):
        nseg, world = int(rng.integers(8, 400)), int(rng.integers(1, 9))
        n = int(rng.integers(2, 200_000))
        rows = rng.integers(0, 1 << 24, size=n, dtype=np.uint64)
        cols = rng.integers(0, nseg * synth.SEGMENT, size=n, dtype=np.uint64)
        recs = oracle.device_order(np.unique(oracle.pack_hits(rows, cols)))
        n = recs.size
        for _ in range(int(rng.integers(0, 6))):                                    # damage: copy a record somewhere else
            a, b = rng.integers(0, n, size=2)
            recs[a] = recs[b]
        spans = [shard_columns(nseg * synth.SEGMENT, r, world) for r in range(world)]
        r_rows, r_cols = oracle.unpack_hits(recs)
        seg = r_cols // np.uint64(synth.SEGMENT)
        key = (seg << np.uint64(38)) | (r_rows.astype(np.uint64) << np.uint64(14)) | (r_cols % np.uint64(synth.SEGMENT))
        bad_order = (np.nonzero(key[1:] <= key[:-1])[0] + 1).tolist()
        # rank stretches: cut the list at random places
        cuts = np.sort(rng.integers(0, n + 1, size=world - 1)).tolist()
        counts = [b - a for a, b in zip([0] + cuts, cuts + [n])]
        owner = np.repeat(np.arange(world), counts)
        lo = np.array([s[0] for s in spans], dtype=np.uint64)[owner]
        hi = np.array([s[1] for s in spans], dtype=np.uint64)[owner]
        bad_span = np.nonzero((r_cols < lo) | (r_cols >= hi))[0]
        t = torch.from_numpy(recs.view(np.int64).copy()).to(dev)
        rep = check_order(t.data_ptr(), n, counts, spans)
        assert rep["records"] == n and rep["out_of_order"] == len(bad_order) and rep["out_of_span"] == bad_span.size, trial
        assert rep["first_out_of_order"] == (bad_order[0] if bad_order else None)
        assert rep["first_out_of_span"] == (int(bad_span[0]) if bad_span.size else None)
    empty = check_order(0, 0)
    assert empty["records"] == 0 and empty["out_of_order"] == 0
