"""The CPU checker against the reference's known answers and golden fixtures (no GPU)."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden_names, load_golden
from havac_amd import synth


def _hits(oracle, symbols, model):
    rows, cols = oracle.unpack_hits(oracle.row_major(oracle.ssv(np.array(symbols, np.uint8), np.array(model, np.int8))))
    return list(zip(rows.tolist(), cols.tolist()))


# Three hand-checked answers recorded from the compiled reference, SURVEY.md App. C.
def test_kat1_three_rows_of_127(oracle):
    assert _hits(oracle, [0] * 5, [[127] * 4] * 3) == [(2, 2), (2, 3), (2, 4)]


def test_kat2_reset_then_threshold(oracle):
    model = [[-128, c, -128, -128] for c in (100, -128, 100, 100, 100)]
    assert _hits(oracle, [1] * 8, model) == [(4, s) for s in range(2, 8)]


def test_kat3_256_hits_255_does_not(oracle):
    model = [[127] * 4, [127] * 4, [2, 1, 0, -1]]
    assert _hits(oracle, [0, 0, 0, 1, 2, 3], model) == [(2, 2)]


def test_cell_forms_agree_on_all_inputs(oracle):
    """SoftSsv.cpp:36-47 (arithmetic) == HavacHls.cpp:376-386 (9-bit carry) for every (prev, match)."""
    L = oracle.lib()
    h1, h2 = C.c_int(0), C.c_int(0)
    for prev in range(256):
        for bits in range(256):
            m = bits - 256 if bits > 127 else bits
            a = L.havac_oracle_cell(prev, m, C.byref(h1))
            b = L.havac_oracle_cell_carry(prev, bits, C.byref(h2))
            assert (a, h1.value) == (b, h2.value), (prev, m)
            t = prev + m
            assert a == (0 if (t < 0 or t >= 256) else t) and h1.value == int(t >= 256)


def test_hit_record_layout(oracle):
    """device/HitReporting.cpp:421-430 and host/Havac.cpp:155-163."""
    rec = oracle.lib().havac_oracle_pack_hit(5, 12288 * 3 + 17)
    assert rec == (17 | (3 << 14) | (5 << 40))
    rows, cols = oracle.unpack_hits(np.array([rec], np.uint64))
    assert (int(rows[0]), int(cols[0])) == (5, 12288 * 3 + 17)
    assert np.array_equal(oracle.pack_hits([5], [12288 * 3 + 17]), np.array([rec], np.uint64))


def test_pack_unpack_2bit(oracle):
    rng = np.random.default_rng(3)
    sym = rng.integers(0, 4, size=4096, dtype=np.uint8)
    packed = oracle.pack_2bit(sym)
    assert packed[0] == sym[0] | (sym[1] << 2) | (sym[2] << 4) | (sym[3] << 6)
    assert np.array_equal(oracle.unpack_2bit(packed), sym)
    assert np.array_equal(synth.pack_2bit(sym), packed)
    assert np.array_equal(synth.unpack_2bit(packed), sym)


def test_device_order(oracle):
    recs = oracle.pack_hits([9, 1, 1, 0], [5, 12288 + 1, 7, 2 * 12288])
    rows, cols = oracle.unpack_hits(oracle.device_order(recs))
    assert list(zip(rows.tolist(), cols.tolist())) == [(1, 7), (9, 5), (1, 12289), (0, 24576)]


@pytest.mark.parametrize("name", golden_names())
def test_golden_fixture(oracle, name):
    packed, model, hits = load_golden(name)
    sym = oracle.unpack_2bit(packed)
    assert np.array_equal(oracle.ssv(sym, model), hits)
    assert np.array_equal(oracle.ssv_mt(sym, model, nthreads=3), hits)
    assert np.array_equal(oracle.ssv_fast(sym, model, nthreads=3), hits)


@pytest.mark.parametrize("name", ["g2_L1024_3seg_cross", "g5d_random_full_range"])
def test_window_equals_slice_of_whole(oracle, name):
    packed, model, hits = load_golden(name)
    sym = oracle.unpack_2bit(packed)
    _, cols = oracle.unpack_hits(hits)
    for lo, hi in ((0, 100), (12000, 12600), (5000, sym.size), (sym.size - 1, sym.size)):
        assert np.array_equal(oracle.ssv_window(sym, model, lo, hi), hits[(cols >= lo) & (cols < hi)])


def test_empty_inputs(oracle):
    assert oracle.ssv(np.zeros(0, np.uint8), np.zeros((4, 4), np.int8)).size == 0
    assert oracle.ssv(np.zeros(16, np.uint8), np.zeros((0, 4), np.int8)).size == 0


def test_restatement_equals_reference_when_built(oracle):
    """Only where oracle/_ref exists (the build container): random sweeps vs the reference object."""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not present on this box")
    rng = np.random.default_rng(11)
    for _ in range(12):
        nrows, n = int(rng.integers(1, 300)), int(rng.integers(1, 4000))
        model = rng.integers(-128, 128, size=(nrows, 4)).astype(np.int8)
        sym = rng.integers(0, 4, size=n, dtype=np.uint8)
        assert np.array_equal(oracle.ssv(sym, model), oracle.ssv_reference(sym, model))
    # the reference's own emission order: rows ascending, columns descending (SoftSsv.cpp:31-32)
    rows, cols = oracle.ssv_reference_raw(np.zeros(5, np.uint8), np.full((3, 4), 127, np.int8))
    assert list(zip(rows.tolist(), cols.tolist())) == [(2, 4), (2, 3), (2, 2)]


def test_fast_route_equals_plain_sweep(oracle):
    """havac_oracle_ssv_fast (AVX2 tiles + threads) is a second route to the same records: ragged sizes around its
    8192-column tile and 16-cell vector, models taller than a tile is wide, extreme scores."""
    rng = np.random.default_rng(21)
    shapes = [(1, 1), (1, 17), (5, 15), (300, 8191), (300, 8192), (300, 8193), (9000, 20_000), (1024, 70_001), (33, 16_400)]
    for k, (nrows, n) in enumerate(shapes):
        if k % 3 == 0:
            model = rng.integers(-128, 128, size=(nrows, 4)).astype(np.int8)
        elif k % 3 == 1:
            model = synth.dfam_like_model(nrows, 50 + k)[0]
        else:
            model = rng.choice(np.array([-128, 127, 126, 1, 0], np.int8), size=(nrows, 4))
        sym = rng.integers(0, 4, size=n, dtype=np.uint8)
        want = oracle.ssv(sym, model)
        for threads in (1, 5):
            assert np.array_equal(oracle.ssv_fast(sym, model, nthreads=threads), want), (nrows, n, threads)
    if oracle.ref_available():
        model = np.full((40, 4), 127, np.int8)
        sym = rng.integers(0, 4, size=30_000, dtype=np.uint8)
        assert np.array_equal(oracle.ssv_fast(sym, model), oracle.ssv_reference(sym, model))


def test_cell_records_follow_the_recurrence(oracle):
    """oracle.cells (the checker of the per-cell trace, test/softSsv/SoftSsv.cpp:59-65): every record is one application of
    the single-cell rule to the record above-left of it, and its crossings are exactly the hit list's"""
    from havac_amd import synth
    model, cons = synth.dfam_like_model(90, 31)
    sym = synth.random_symbols(4000, 32)
    synth.plant_homologs(sym, cons, sym.size, every=600, length=80, sub=0.05)
    row0, col0, h, w = 10, 100, 80, 700
    cells = oracle.cells(sym, model, row0, col0, h, w)
    assert cells["written"].all() and not cells["pending"].any()
    assert np.array_equal(cells["symbol"], np.broadcast_to(sym[col0:col0 + w], (h, w)))
    rows = np.arange(row0, row0 + h)[:, None]
    assert np.array_equal(cells["match"], model[rows, cells["symbol"]])
    assert np.array_equal(cells["prev"][1:, 1:], cells["score"][:-1, :-1])      # the diagonal's previous cell
    t = cells["prev"].astype(int) + cells["match"].astype(int)
    assert np.array_equal(cells["hit"], (t >= 256).astype(np.uint8))
    assert np.array_equal(cells["score"], np.where((t < 0) | (t >= 256), 0, t).astype(np.uint8))
    r, c = oracle.unpack_hits(oracle.ssv(sym, model))
    inside = (r >= row0) & (r < row0 + h) & (c >= col0) & (c < col0 + w)
    want = np.zeros((h, w), np.uint8)
    want[(r[inside] - row0).astype(int), (c[inside] - col0).astype(int)] = 1
    assert np.array_equal(cells["hit"], want)
