"""Per-cell differential test: the traced instantiation of the SSV kernel against the CPU checker, cell by cell.

The reference's counterpart is its HAVAC_PER_CELL_DATA_TESTING build (device/PublicDefines.h:11): every cell processor
records {prevValue, matchScore, cellValue, symbol, passesThreshold} (device/HavacHls.cpp:388-399), softSsv does the same
(test/softSsv/SoftSsv.cpp:59-65) and test/byCellComparator/byCellComparator.cpp:47-96 requires every cell to be present
on both sides and equal.  Here: havac_ssv_set_cell_trace + oracle.cells, through the C ABI (GPU only)."""
import numpy as np
import pytest
import torch

from havac_amd import synth
from havac_amd.ssv import SsvContext

pytestmark = pytest.mark.gpu

FIELDS = ("prev", "match", "score", "hit", "symbol")


def traced_pass(packed, model, window, hit_capacity=1 << 20):
    """one pass with the trace window (row0, col0, h, w) -> (cell records [h, w], ordered hit records)"""
    from oracle import pyoracle as O
    dev = torch.device("cuda", 0)
    row0, col0, h, w = window
    d_seq = torch.from_numpy(packed).to(dev)
    d_phmm = torch.from_numpy(np.ascontiguousarray(model).reshape(-1)).to(dev)
    d_hits = torch.empty(hit_capacity, dtype=torch.int64, device=dev)
    d_cells = torch.zeros(h * w * 8, dtype=torch.uint8, device=dev)
    ctx = SsvContext()
    try:
        ctx.set_cell_trace(d_cells.data_ptr(), row0, col0, h, w)
        torch.cuda.synchronize()
        ctx.enqueue(d_seq.data_ptr(), packed.size * 4, d_phmm.data_ptr(), model.shape[0], d_hits.data_ptr(), hit_capacity)
        found = ctx.finish()
        ctx.set_cell_trace(0)
        cells = d_cells.cpu().numpy().view(O.CELL_RECORD).reshape(h, w)
        hits = d_hits[:found].cpu().numpy().view(np.uint64)
        # the same context, trace off again: the production kernel gives the same list
        ctx.enqueue(d_seq.data_ptr(), packed.size * 4, d_phmm.data_ptr(), model.shape[0], d_hits.data_ptr(), hit_capacity)
        found_again = ctx.finish()
        assert np.array_equal(d_hits[:found_again].cpu().numpy().view(np.uint64), hits)
    finally:
        ctx.close()
    return cells, hits


def compare(cells, want, oracle_hits_in_window):
    """every cell present; every cell that is not `pending` equal in all five fields; pending cells only below a crossing"""
    assert cells["written"].all(), f"{int((cells['written'] == 0).sum())} cells of the window were never visited"
    assert (cells["zero"] == 0).all()
    settled = cells["pending"] == 0
    for f in FIELDS:
        bad = np.argwhere(settled & (cells[f] != want[f]))
        assert bad.size == 0, (f"field {f}: {bad.shape[0]} cells differ, first at window (row, col) {tuple(bad[0])}: "
                               f"GPU {cells[tuple(bad[0])]} checker {want[tuple(bad[0])]}")
    # symbol and match score do not depend on the score: they must be right under `pending` too
    for f in ("match", "symbol"):
        assert np.array_equal(cells[f], want[f]), f"field {f} differs in a pending cell"
    # a pending cell lies 1..3 steps below a crossing on its diagonal (or the crossing is above / left of the window)
    pend = np.argwhere(cells["pending"] == 1)
    for r, c in pend:
        above = [(r - k, c - k) for k in (1, 2, 3) if r - k >= 0 and c - k >= 0]
        assert len(above) < 3 or any(want["hit"][a] for a in above), f"pending cell ({r}, {c}) with no crossing above it"
    assert int(want["hit"].sum()) == oracle_hits_in_window
    return int(pend.shape[0])


@pytest.mark.parametrize("nrows,nseg,seed,window", [
    (100, 1, 1, (0, 0, 64, 64)),              # the matrix's corner: column 0 and row 0 (diagonals that enter at the edge)
    (100, 1, 2, (36, 6000, 64, 64)),          # the model's last rows: the step behind the last chunk (high cells one row behind)
    (100, 2, 3, (20, 12288 - 32, 64, 64)),    # across a segment boundary
    (100, 1, 4, (0, 12288 - 64, 100, 64)),    # the last columns, the whole height
    (1024, 2, 5, (480, 2048 - 40, 64, 80)),   # across a tile boundary (2048 diagonals) and a chunk boundary
    (33, 1, 6, (0, 500, 33, 64)),             # one row into the second chunk
])
def test_cells_equal_the_checker(oracle, nrows, nseg, seed, window):
    model, cons = synth.dfam_like_model(nrows, 100 + seed)
    sym = synth.random_symbols(nseg * synth.SEGMENT, 200 + seed)
    synth.plant_homologs(sym, cons, sym.size, every=700, length=min(nrows, 120), sub=0.1)
    row0, col0, h, w = window
    cells, hits = traced_pass(synth.pack_2bit(sym), model, window)
    assert np.array_equal(hits, oracle.ssv(sym, model)), "the traced kernel's hit list differs"
    want = oracle.cells(sym, model, row0, col0, h, w)
    r, c = oracle.unpack_hits(hits)
    inside = int(((r >= row0) & (r < row0 + h) & (c >= col0) & (c < col0 + w)).sum())
    compare(cells, want, inside)


def test_cells_around_crossings(oracle):
    """windows laid over hits: the crossing itself is exact, the cells behind it inside the same four steps are pending"""
    nrows = 200
    model, cons = synth.dfam_like_model(nrows, 77)
    sym = synth.random_symbols(synth.SEGMENT, 78)
    synth.plant_homologs(sym, cons, sym.size, every=500, length=150, sub=0.05)
    hits = oracle.ssv(sym, model)
    assert hits.size > 10
    rows, cols = oracle.unpack_hits(hits)
    seen_pending = 0
    for k in np.linspace(0, hits.size - 1, 6).astype(int):
        row0 = int(min(max(int(rows[k]) - 32, 0), nrows - 64))
        col0 = int(min(max(int(cols[k]) - 32, 0), sym.size - 64))
        cells, got = traced_pass(synth.pack_2bit(sym), model, (row0, col0, 64, 64))
        assert np.array_equal(got, hits)
        want = oracle.cells(sym, model, row0, col0, 64, 64)
        inside = int(((rows >= row0) & (rows < row0 + 64) & (cols >= col0) & (cols < col0 + 64)).sum())
        assert inside >= 1
        seen_pending += compare(cells, want, inside)
        assert cells["hit"][int(rows[k]) - row0, int(cols[k]) - col0] == 1
    assert seen_pending > 0        # a crossing in steps 0..2 of a window leaves pending cells: the flag is exercised


def test_cells_with_full_range_scores(oracle):
    """a model with scores over the whole int8 range: chunks that test every two steps, both saturations"""
    rng = np.random.default_rng(5)
    nrows = 96
    model = rng.integers(-128, 128, size=(nrows, 4), dtype=np.int8)
    sym = synth.random_symbols(synth.SEGMENT, 9)
    window = (16, 3000, 64, 64)
    cells, hits = traced_pass(synth.pack_2bit(sym), model, window, hit_capacity=1 << 22)
    assert np.array_equal(hits, oracle.ssv(sym, model, cap=1 << 22))
    want = oracle.cells(sym, model, *window)
    r, c = oracle.unpack_hits(hits)
    inside = int(((r >= 16) & (r < 80) & (c >= 3000) & (c < 3064)).sum())
    compare(cells, want, inside)
