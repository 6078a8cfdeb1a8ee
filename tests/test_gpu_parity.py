"""Parity of the HIP path against the CPU checker, through the C ABI (GPU only)."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from havac_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def client():
    from havac_amd.hw_client import HavacHwClient
    c = HavacHwClient(deviceIndex=0)
    yield c
    c.close()


def run(client, packed, model):
    client.writeSequence(packed)
    client.writePhmm(model)
    client.invokeHavacSsvAsync()
    assert client.waitForHavacSsvAsync() == 4      # HAVAC_CMD_STATE_COMPLETED
    return client.getHitList()


def first_difference(got, want, oracle):
    n = min(got.size, want.size)
    bad = np.nonzero(got[:n] != want[:n])[0]
    k = int(bad[0]) if bad.size else n
    def show(a):
        if k >= a.size:
            return "<end>"
        r, c = oracle.unpack_hits(a[k:k + 1])
        return f"(row {int(r[0])}, col {int(c[0])})"
    return f"got {got.size} hits, want {want.size}; first difference at #{k}: got {show(got)} want {show(want)}"


@pytest.mark.parametrize("name", golden_names())
def test_golden_fixture(client, oracle, name):
    packed, model, hits = load_golden(name)
    client.setHitCapacity(max(1 << 20, hits.size + 16))
    got = run(client, packed, model)
    assert np.array_equal(got, hits), first_difference(got, hits, oracle)


@pytest.mark.parametrize("nrows,nseg,kind,seed", [
    (1, 1, "dfam", 1), (2, 1, "rand", 2), (31, 1, "rand", 3), (32, 1, "rand", 4), (33, 2, "rand", 5),
    (100, 1, "dfam", 6), (1024, 4, "dfam", 7), (2047, 3, "dfam", 8), (2048, 3, "rand", 9),
    (2100, 2, "dfam", 10), (5000, 1, "dfam", 11), (777, 9, "mixed", 12), (64, 40, "dfam", 13),
])
def test_random_against_oracle(client, oracle, nrows, nseg, kind, seed):
    rng = np.random.default_rng(seed)
    n = nseg * synth.SEGMENT
    sym = synth.random_symbols(n - int(rng.integers(0, 3000)), seed=100 + seed)
    if kind == "rand":
        model = rng.integers(-128, 128, size=(nrows, 4)).astype(np.int8)
    else:
        model, cons = synth.dfam_like_model(nrows, 200 + seed)
        synth.plant_homologs(sym, cons, n, every=7001, length=min(nrows, 300), seed=300 + seed)
        if kind == "mixed":
            model[rng.integers(0, nrows, size=nrows // 10)] = 127
    want = oracle.ssv_mt(sym, model)
    client.setHitCapacity(max(1 << 20, want.size + 16))
    got = run(client, synth.pack_2bit(sym), model)
    assert np.array_equal(got, want), first_difference(got, want, oracle)


@pytest.mark.parametrize("seed", range(6))
def test_extreme_models(client, oracle, seed):
    """Scores at the int8 extremes and crossings on consecutive rows: both steps of a pair, both cells of a
    register, repairs after a first-step crossing, saturation back down to zero."""
    rng = np.random.default_rng(1000 + seed)
    nrows = int(rng.integers(33, 200))
    choices = [np.array([-128, 127]), np.array([-128, -1, 0, 1, 127]), np.array([126, 127, -127, -128]),
               np.array([64, 127, -64]), np.array([127]), np.array([-128, 127, 127, 127])][seed]
    model = rng.choice(choices, size=(nrows, 4)).astype(np.int8)
    sym = synth.random_symbols(2 * synth.SEGMENT - 77, seed=2000 + seed)
    want = oracle.ssv_mt(sym, model)
    client.setHitCapacity(max(1 << 20, want.size + 16))
    got = run(client, synth.pack_2bit(sym), model)
    assert got.size == want.size and np.array_equal(got, want), first_difference(got, want, oracle)


@pytest.mark.parametrize("nrows", [150, 700])
def test_marks_that_would_vanish_inside_a_window(client, oracle, nrows):
    """Round 5: whether a four-step window may look for hits at its end only is decided per WINDOW (a byte per 32-row chunk,
    ssv_prepare_model).  A mild model -- every window safe -- with, at every residue of the row index modulo 32 in turn, a crossing
    forced by rows of +127 and then two or three rows that take the mark away again within the window if nobody looks in its middle
    (-128 or -90 for every symbol): the windows those rows lie in must test twice, their neighbours need not.  The resident-table
    kernel (150 rows) and the standard one (700 rows, and 150 rows forced), sequences with and without planted homologs."""
    rng = np.random.default_rng(nrows)
    model = rng.integers(-40, -9, size=(nrows, 4)).astype(np.int8)
    cons = rng.integers(0, 4, size=nrows)
    model[np.arange(nrows), cons] = rng.integers(30, 60, size=nrows).astype(np.int8)
    # the bait: from row r on, three rows of +127 for every symbol (a sure crossing on rows r+1 or r+2 wherever the score is above
    # 2), then `k` rows that score `low` for every symbol; r walks through every residue modulo 32 and both cells' phases
    r, step = 3, 37
    while r + 6 < nrows:
        k, low = int(rng.integers(2, 4)), int(rng.choice([-128, -90]))
        model[r:r + 3] = 127
        model[r + 3:r + 3 + k] = low
        r += step
    n = 3 * synth.SEGMENT
    sym = synth.random_symbols(n, seed=nrows + 1)
    synth.plant_homologs(sym, cons.astype(np.uint8), n, every=2500, length=min(nrows, 300), sub=0.02, seed=nrows + 2)
    want = oracle.ssv_mt(sym, model)
    assert want.size > 10_000
    client.setHitCapacity(want.size + 64)
    for variant in ([-1] * 9, [-1] * 8 + [0]):
        client.setTuning(*variant)
        got = run(client, synth.pack_2bit(sym), model)
        assert got.size == want.size and np.array_equal(got, want), (variant, first_difference(got, want, oracle))
    client.setTuning(*([-1] * 9))


@pytest.mark.parametrize("nrows", [1, 7, 16, 17, 33, 40, 48, 49, 80, 100, 144, 145, 240])
def test_models_that_end_in_the_first_half_of_their_last_chunk(client, oracle, nrows):
    """Round 5: the resident-table kernel does not run the second half of a model's last chunk where the model's rows end in the
    first (rows mod 32 in 1 .. 16: sixteen steps of padding rows that change no score; the step the lagging cells still owe is
    taken in the chunk's middle).  Heights on both sides of the rule, the model's last rows deciding hits, every kernel."""
    rng = np.random.default_rng(5000 + nrows)
    model = rng.integers(-40, 60, size=(nrows, 4)).astype(np.int8)
    model[-1] = 127                                       # the very last row decides hits: the lagging cells' last step matters
    if nrows > 2:
        model[-2] = rng.integers(60, 128, size=4).astype(np.int8)
    sym = synth.random_symbols(2 * synth.SEGMENT, seed=6000 + nrows)
    want = oracle.ssv_mt(sym, model)
    assert want.size > 100 or nrows < 3                   # (one or two rows cannot reach 256)
    client.setHitCapacity(want.size + 64)
    for variant in ([-1] * 9, [-1] * 8 + [0], [-1, 2, -1, -1, -1, -1, -1, -1, 1]):
        client.setTuning(*variant)
        got = run(client, synth.pack_2bit(sym), model)
        assert got.size == want.size and np.array_equal(got, want), (variant, first_difference(got, want, oracle))
    client.setTuning(*([-1] * 9))


def test_rows_exactly_at_chunk_multiples(client, oracle):
    """Model lengths 32k-1, 32k, 32k+1 around the kernel's 32-row chunk: the last row belongs to the lagging cells."""
    rng = np.random.default_rng(77)
    sym = synth.random_symbols(synth.SEGMENT, seed=78)
    for nrows in (31, 32, 33, 63, 64, 65, 96):
        model = rng.integers(-40, 128, size=(nrows, 4)).astype(np.int8)
        model[-1] = 127                                   # make the very last row decide hits
        want = oracle.ssv(sym, model)
        client.setHitCapacity(max(1 << 20, want.size + 16))
        got = run(client, synth.pack_2bit(sym), model)
        assert np.array_equal(got, want), (nrows, first_difference(got, want, oracle))
