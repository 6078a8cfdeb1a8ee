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
