"""Generates tests/golden/g8_boundary_*.npz: what the REFERENCE's per-(model, record) CPU SSV returns on small files.

Run in the build container only (needs /root/reference):

    make -C oracle all _ref && make -C tests/refhost && havac_amd/csrc/host/build.sh && python tests/golden/make_golden_g8.py

The answers come from HitsFromSsv (host/test/Ssv.cpp:8-68) -- the function the reference's on-FPGA test compares the
device with (host/test/RefernceComparisonTest/ReferenceComparisonTest.cpp:52-128) and the one boundary mode (SURVEY.md
section 8 row f2) claims to match -- compiled from where it lies by tests/refhost/Makefile against the product's reader
headers (the reference's reader libraries are un-vendored: a differential build, like G7's).

Each fixture is data only: the FASTA text, the HMMER3/f text, the p-value, and HitsFromSsv's hits as rows
(sequenceNumber, phmmNumber, sequencePosition, phmmPosition) in its emission order.

One restriction, checked here for every model of every case and recorded in the fixture: HitsFromSsv projects scores with
the legacy per-score function emissionScoreToProjectedScore (PhmmReprojection.cpp:88-107: -log2e * (s - 2/log2e) * m),
the device path with the table loop of p7HmmProjectForThreshold256 (:133-143: 2m - s * (log2e * m)).  The two are
different float expressions and can round a value next to a .5 boundary differently (G7 records both).  A model on which
they differ anywhere is not used (its seed is bumped): on such a model the reference's two CPU implementations
themselves disagree, and there is no single reference answer to pin.
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.dirname(os.path.abspath(__file__))

from havac_amd import havac, synth            # noqa: E402
from oracle import pyoracle as O              # noqa: E402
from refhost import binding                   # noqa: E402
from make_golden_g7 import load as load_reproj, through_text   # noqa: E402

REPROJ = load_reproj("libreprojection_ref_O0.so")
TMP = tempfile.mkdtemp(prefix="g8_")


def model_where_both_projections_agree(L, seed, p, maxl=None, mu=-9.2, lam=0.71, star=False):
    """-> (dict for synth.write_hmm, consensus, int8 table) with single == table on every entry"""
    while True:
        _, cons = synth.dfam_like_model(L, seed)
        em = synth.emissions_from_consensus(cons, seed + 1000)
        if star:
            rng = np.random.default_rng(seed)
            mask = rng.random(em.shape) < 0.08
            mask[np.arange(L), cons] = False
            em[mask] = np.inf
        em32 = through_text(np.where(np.isfinite(em), em, 0.0), "%.5f")
        em32[~np.isfinite(em)] = np.inf
        mu32, lam32, p32 = np.float32("%.4f" % mu), np.float32("%.5f" % lam), np.float32(p)
        maxl_ = maxl or 3 * L + 50
        scale = np.float32(REPROJ.reproj_ref_scale(mu32, lam32, maxl_, L, p32))
        table = np.empty((L, 4), np.int8)
        REPROJ.reproj_ref_project(mu32, lam32, maxl_, L, p32, em32.ctypes.data, table.ctypes.data)
        single = np.array([REPROJ.reproj_ref_score(v, scale) for v in em32.ravel()], np.float32).reshape(L, 4)
        if np.array_equal(single.astype(np.int8), table):
            return dict(name=f"fam{seed}", acc=f"RF{seed:05d}", emissions=em, maxl=maxl_, mu=mu, lam=lam), cons, table
        seed += 100000          # the two expressions disagree somewhere on this model: not usable, see the docstring


def text_of(symbols):
    return "".join("ACGT"[v] for v in symbols)


def mutate(piece, rng, sub=0.1):
    piece = piece.copy()
    m = rng.random(piece.size) < sub
    piece[m] = rng.integers(0, 4, size=int(m.sum()))
    return piece


def reference_hits(name, records, models, p):
    fa, hmm = os.path.join(TMP, name + ".fa"), os.path.join(TMP, name + ".hmm")
    synth.write_fasta(fa, records)
    synth.write_hmm(hmm, models)
    return binding.reference_ssv_hits(fa, hmm, p), open(fa).read(), open(hmm).read(), fa, hmm


def default_mode_hits(records, tables):
    """what the DEVICE semantics (one diagonal field over the concatenation) give on the same inputs, with
    every terminator and non-acg character as T: only used to show that a case really has a diagonal that crosses"""
    lut = np.full(256, 3, np.uint8)
    for ch, v in zip(b"ACGacg", [0, 1, 2, 0, 1, 2]):
        lut[ch] = v
    sym = np.concatenate([np.concatenate([lut[np.frombuffer(t.encode(), np.uint8)], [3]]) for _, t in records]).astype(np.uint8)
    pad = synth.padded_length(sym.size)
    sym = np.concatenate([sym, np.zeros(pad - sym.size, np.uint8)])
    rows, cols = O.unpack_hits(O.ssv(sym, np.concatenate(tables)))
    ends = np.cumsum([len(t) + 1 for _, t in records])
    mstarts = np.concatenate([[0], np.cumsum([t.shape[0] for t in tables])])
    out = set()
    for r, c in zip(rows.tolist(), cols.tolist()):
        if c >= ends[-1]:
            continue
        j = int(np.searchsorted(ends, c, side="right"))
        k = int(np.searchsorted(mstarts, r, side="right")) - 1
        out.add((j, k, c - (int(ends[j - 1]) if j else 0), r - int(mstarts[k])))
    return out


def save(name, p, fasta_text, hmm_text, hits, note, **extra):
    np.savez_compressed(os.path.join(OUT, f"g8_boundary_{name}.npz"), fasta=np.array(fasta_text), hmm=np.array(hmm_text),
                        p=np.float32(p), hits=hits.astype(np.uint32), single_equals_table=np.array(True),
                        note=np.array(note), **extra)
    print(f"g8_boundary_{name}: {hits.shape[0]} hits, {len(fasta_text)} B fasta, {len(hmm_text)} B hmm; {note}")


def main():
    rng = np.random.default_rng(8008)

    # ---- 1. several models x several records ------------------------------------------------------------------
    p = 0.02
    ms = [model_where_both_projections_agree(L, 10 + k, p) for k, L in enumerate([60, 300, 150])]
    cons = np.concatenate([m[1] for m in ms])
    records = []
    for k, n in enumerate([3000, 5000, 800, 17]):
        s = rng.integers(0, 4, size=n, dtype=np.uint8)
        synth.plant_homologs(s, cons, n, every=700, length=min(220, n), seed=k)
        records.append((f"seq{k}", text_of(s)))
    hits, fat, hmt, *_ = reference_hits("multi", records, [m[0] for m in ms], p)
    assert hits.shape[0] > 40 and len(set(hits[:, 0])) >= 3 and len(set(hits[:, 1])) == 3
    save("multi", p, fat, hmt, hits, "3 models x 4 records, planted homologs")

    # ---- 2. everything that is not a/c/g is T (host/test/Ssv.cpp:29-34), '*' emissions, p = 1e-4 --------------------
    p = 1e-4
    ms = [model_where_both_projections_agree(L, 20 + k, p, star=True) for k, L in enumerate([120, 90])]
    cons = np.concatenate([m[1] for m in ms])
    records = []
    for k, n in enumerate([2500, 1800]):
        s = rng.integers(0, 4, size=n, dtype=np.uint8)
        synth.plant_homologs(s, cons, n, every=500, length=160, sub=0.05, seed=10 + k)
        t = np.array(list(text_of(s)), dtype="U1")
        t[s == 3] = rng.choice(list("TtNnRYKMSWBDHVXUu*-"), size=int((s == 3).sum()))     # every T as some non-acg character
        lower = rng.random(n) < 0.3
        t[lower] = np.char.lower(t[lower])
        records.append((f"amb{k}", "".join(t)))
    hits, fat, hmt, *_ = reference_hits("nonacgt", records, [m[0] for m in ms], p)
    assert hits.shape[0] > 20
    save("nonacgt", p, fat, hmt, hits, "ambiguity codes / N / lower case / '*' emissions; non-acg = T; p = 1e-4")

    # ---- 3. a hit on a record's last residue, and one on its terminator column ---------------------------------------
    p = 0.02
    m, cons, table = model_where_both_projections_agree(200, 30, p)
    base = rng.integers(0, 4, size=600, dtype=np.uint8)
    found_last = found_term = None
    for trim in range(0, 60):
        body = np.concatenate([base, cons[20:200 - trim]])
        for tail in ("", "A"):                      # with "A" behind it the stretch ends one before the last residue
            recs = [("r0", text_of(body) + tail), ("r1", text_of(rng.integers(0, 4, size=300, dtype=np.uint8)))]
            h, fat, hmt, *_ = reference_hits("edge", recs, [m], p)
            n0 = len(recs[0][1])
            on_last = [x for x in h if x[0] == 0 and x[2] == n0 - 1]
            on_term = [x for x in h if x[0] == 0 and x[2] == n0]
            if on_last and found_last is None:
                found_last = (recs, h, fat, hmt)
            if on_term and found_term is None:
                found_term = (recs, h, fat, hmt)
        if found_last and found_term:
            break
    assert found_last is not None, "no case with a hit on the last residue"
    recs, h, fat, hmt = found_last
    save("last_residue", p, fat, hmt, h, "a hit at sequencePosition == record length - 1")
    if found_term is not None:
        recs, h, fat, hmt = found_term
        save("terminator", p, fat, hmt, h, "a hit on the record's terminator column (scored as T): sequencePosition == record length")

    # ---- 4. a diagonal that would cross a RECORD boundary ----------------------------------------------------------
    m, cons, table = model_where_both_projections_agree(200, 40, p)
    for split in range(100, 140):
        a = np.concatenate([rng.integers(0, 4, size=400, dtype=np.uint8), cons[0:split]])
        b = np.concatenate([cons[split + 1:200], rng.integers(0, 4, size=400, dtype=np.uint8)])    # the terminator takes row `split`
        recs = [("left", text_of(a)), ("right", text_of(b))]
        h, fat, hmt, *_ = reference_hits("xrec", recs, [m], p)
        ref = {tuple(int(v) for v in x) for x in h}
        crossing = sorted(x for x in default_mode_hits(recs, [table]) - ref if x[0] == 1 and x[2] < 8)
        if crossing:
            break
    assert crossing, "no diagonal crosses the record boundary"
    save("cross_record", p, fat, hmt, h, "a homolog straddles two records: the device's default mode hits early in record 1, "
         "HitsFromSsv does not", default_mode_only=np.array(crossing, np.uint32))

    # ---- 5. a diagonal that would cross a MODEL boundary -----------------------------------------------------------
    ms = [model_where_both_projections_agree(L, 50 + k, p) for k, L in enumerate([60, 80])]
    for keep in range(20, 50):
        piece = np.concatenate([ms[0][1][60 - keep:], ms[1][1][:40]])      # model 0's last rows, then model 1's first rows
        s = np.concatenate([rng.integers(0, 4, size=500, dtype=np.uint8), piece, rng.integers(0, 4, size=500, dtype=np.uint8)])
        recs = [("only", text_of(s))]
        h, fat, hmt, *_ = reference_hits("xmod", recs, [x[0] for x in ms], p)
        ref = {tuple(int(v) for v in x) for x in h}
        crossing = sorted(x for x in default_mode_hits(recs, [x[2] for x in ms]) - ref if x[1] == 1 and x[3] < 8)
        if crossing:
            break
    assert crossing, "no diagonal crosses the model boundary"
    save("cross_model", p, fat, hmt, h, "a stretch matches model 0's end and model 1's start: the default mode hits in model 1's "
         "first rows, HitsFromSsv does not", default_mode_only=np.array(crossing, np.uint32))

    # ---- 6. ragged: empty record, one residue, records shorter than the models, a model of one row ---------------
    ms = [model_where_both_projections_agree(L, 60 + k, p) for k, L in enumerate([1, 250, 40])]
    cons = np.concatenate([m[1] for m in ms])
    lens = [0, 1, 30, 2000, 2, 249, 1500]
    records = []
    for k, n in enumerate(lens):
        s = rng.integers(0, 4, size=n, dtype=np.uint8)
        if n >= 30:
            synth.plant_homologs(s, cons, n, every=max(60, n // 4), length=min(240, n), seed=20 + k)
        records.append((f"rag{k}", text_of(s)))
    hits, fat, hmt, *_ = reference_hits("ragged", records, [m[0] for m in ms], p)
    assert hits.shape[0] > 10
    save("ragged", p, fat, hmt, hits, "records of 0, 1, 2, 30, 249 residues against models of 1, 250, 40 rows")


if __name__ == "__main__":
    main()
