"""Boundary mode against the function it imitates: tests/golden/g8_boundary_*.npz hold what
the REFERENCE's HitsFromSsv (host/test/Ssv.cpp:8-68) returned on small FASTA / .hmm files, captured in the build container
by tests/golden/make_golden_g8.py from the reference's own code (the function its on-FPGA test compares the device with,
host/test/RefernceComparisonTest/ReferenceComparisonTest.cpp:52-128).

Here: the CPU half (the product's reader + projection and the checker's per-pair sweep reproduce the fixture exactly)
and the GPU half (`Havac` in boundary mode returns exactly the fixture's hits, on the GPU-built and on the host-built
layout).  Models are restricted to those on which the reference's two projection expressions agree (see the generator)."""
import numpy as np
import pytest

from conftest import g8_names, load_g8

NAMES = g8_names()


def test_there_are_at_least_six_fixtures_and_they_cover_the_named_cases():
    assert len(NAMES) >= 6
    for needed in ("multi", "nonacgt", "last_residue", "cross_record", "cross_model", "ragged"):
        assert "g8_boundary_" + needed in NAMES


def records_of(fasta_path):
    records, cur = [], None
    for line in open(fasta_path):
        if line.startswith(">"):
            if cur is not None:
                records.append("".join(cur))
            cur = []
        else:
            cur.append(line.strip())
    records.append("".join(cur))
    return records


@pytest.mark.parametrize("name", NAMES)
def test_per_pair_checker_with_the_product_projection_equals_hits_from_ssv(name, tmp_path, oracle):
    from havac_amd import havac
    fa, hmm, p, want, z = load_g8(name, tmp_path)
    table, lens = havac.project_hmm(hmm, p)                       # host only: the product's reader + table projection
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64).tolist()
    lut = np.full(256, 3, np.uint8)                               # host/test/Ssv.cpp:29-34: a/c/g, everything else T
    for ch, v in zip(b"ACGacg", [0, 1, 2, 0, 1, 2]):
        lut[ch] = v
    got = []
    for j, text in enumerate(records_of(fa)):
        sym = np.concatenate([lut[np.frombuffer(text.encode(), np.uint8)], [3]]).astype(np.uint8)   # + the terminator column
        for k in range(len(lens)):
            r, c = oracle.unpack_hits(oracle.ssv(sym, table[starts[k]:starts[k + 1]]))
            got += [(j, k, int(cc), int(rr)) for rr, cc in zip(r, c)]
    assert sorted(got) == want
    if name.endswith("last_residue"):
        n0 = len(records_of(fa)[0])
        assert any(x[0] == 0 and x[2] == n0 - 1 for x in want)
    if name.endswith("terminator"):
        n0 = len(records_of(fa)[0])
        assert any(x[0] == 0 and x[2] == n0 for x in want)


@pytest.mark.gpu
@pytest.mark.parametrize("device_packing", [True, False])
@pytest.mark.parametrize("name", NAMES)
def test_havac_in_boundary_mode_equals_hits_from_ssv(name, device_packing, tmp_path):
    from havac_amd import havac
    fa, hmm, p, want, z = load_g8(name, tmp_path)
    h = havac.Havac(0, p)
    h.setBoundaryMode(True)
    h.setDevicePacking(device_packing)
    h.loadPhmm(hmm)
    h.loadSequence(fa)
    h.runHardwareClient()
    hits = h.getHitsFromFinishedRun()
    h.close()
    got = sorted((x.sequenceIndex, x.phmmIndex, x.sequencePosition, x.phmmPosition) for x in hits)
    assert got == want                                            # the same multiset of hits, exactly
    if "default_mode_only" in z.files and device_packing:
        # and the case is what it says: under the reference's DEVICE semantics (the default mode) a diagonal crosses the
        # boundary and hits early on the far side, where HitsFromSsv has nothing (the fixture lists those hits for 'T' at
        # the terminator; the default mode draws rand() there, so only the far side's first positions are compared)
        d = havac.Havac(0, p)
        d.loadPhmm(hmm)
        d.loadSequence(fa)
        d.runHardwareClient()
        default = sorted((x.sequenceIndex, x.phmmIndex, x.sequencePosition, x.phmmPosition) for x in d.getHitsFromFinishedRun())
        d.close()
        assert default != want
