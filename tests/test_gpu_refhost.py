"""Drop-in, as a tested fact: the REFERENCE's own host code -- host/Havac.cpp:20-191, host/phmm/PhmmPreprocessor.cpp:9-31,
host/sequence/SequencePreprocessor.cpp:9-84, PhmmReprojection/PhmmReprojection.cpp -- compiled unchanged from where it
lies (tests/refhost/Makefile, build container) against integration/HavacHwClient.hpp, runs on the GPU over the C ABI of
include/havac_dev.h, next to this repository's own `Havac`, on the same files with the same srand().

Differential, not absolute: the FASTA / HMMER3 reader libraries the reference links are un-vendored, so both sides read
the files with this repository's readers.  Everything between the readers and the device -- projection arithmetic,
concatenation, the packer with its rand() order and its 'Y' quirk, the control flow and state checks of Havac, the hit
resolver -- is the reference's code on one side and the product's on the other."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from havac_amd import synth
from refhost import binding
from test_gpu_api import write_inputs

pytestmark = pytest.mark.gpu

if not os.path.isfile(binding.REFHOST_LIB):
    pytest.skip("tests/_refhost/libhavac_refhost.so was not built (it is made in the build container, where the "
                "reference tree is, and travels to the GPU box)", allow_module_level=True)

libc = C.CDLL(None)


def product_tuples(hits):
    return [(x.sequencePosition, x.sequenceIndex, x.phmmPosition, x.phmmIndex) for x in hits]


def run_both(fa, hmm, seed, p=0.02):
    from havac_amd import havac
    ref = binding.ReferenceHavac(0, p)
    ref.loadPhmm(hmm)
    libc.srand(seed)
    ref.loadSequence(fa)
    ref.runHardwareClient()
    assert ref.currentHardwareState() == 4
    ref_hits = ref.getHitsFromFinishedRun()
    ref_strings = [ref.hitToString(i) for i in range(min(len(ref_hits), 50))]
    ref.close()

    ours = havac.Havac(0, p)
    ours.loadPhmm(hmm)
    libc.srand(seed)
    ours.loadSequence(fa)                 # packs on the GPU by default; draws the same rand() values in the same order
    ours.runHardwareClient()
    our_hits = ours.getHitsFromFinishedRun()
    raw = ours.rawHits()
    ours.close()
    return ref_hits, ref_strings, our_hits, raw


@pytest.mark.parametrize("seed", [4242, 7])
def test_reference_havac_and_product_havac_return_the_same_hits(tmp_path, oracle, seed):
    from havac_amd import havac
    fa, hmm = write_inputs(tmp_path, [60, 300, 150], [5000, 9000, 30000, 17], seed=seed)
    ref_hits, ref_strings, our_hits, raw = run_both(fa, hmm, seed)
    assert len(ref_hits) > 20
    assert ref_hits == product_tuples(our_hits)                         # element for element, same order
    assert ref_strings == [x.toString() for x in our_hits[:50]]          # HavacHit::toString, host/Havac.cpp:202-207
    # and both are right: the raw records are the checker's for what the packer and the projection produced
    packed, _, _ = havac.pack_fasta(fa, seed=seed)
    table, _ = havac.project_hmm(hmm, 0.02)
    assert np.array_equal(raw, oracle.ssv(oracle.unpack_2bit(packed), table))


def test_reference_packer_quirks_reach_the_device_identically(tmp_path):
    """Every ambiguity code, lower case, 'Y' (host/sequence/SequencePreprocessor.cpp:77: always A unless last), N runs and
    record terminators draw from rand(): the reference's packer on the host and the product's on-GPU packing + patches
    must leave the same symbols, or the hit lists below differ around the planted homologs."""
    rng = np.random.default_rng(11)
    _, cons = synth.dfam_like_model(120, 70)
    models = [dict(name="fam0", acc="RF00000", emissions=synth.emissions_from_consensus(cons, 80), maxl=400, mu=-9.1, lam=0.71)]
    records = []
    for k, n in enumerate([3000, 1, 2500, 0, 4000]):
        s = rng.integers(0, 4, size=n, dtype=np.uint8)
        synth.plant_homologs(s, cons, n, every=700, length=110, seed=k)
        text = np.array(list("".join("ACGT"[v] for v in s)), dtype="U1")
        if n > 100:
            where = rng.choice(n, size=n // 12, replace=False)
            text[where] = rng.choice(list("NRYKMSWBDHVnrykmswacgtXU*-"), size=where.size)
        records.append((f"r{k}", "".join(text) + ("Y" if k == 4 else "")))
    fa, hmm = str(tmp_path / "q.fa"), str(tmp_path / "q.hmm")
    synth.write_fasta(fa, records)
    synth.write_hmm(hmm, models)
    for seed in (1, 99):
        ref_hits, _, our_hits, _ = run_both(fa, hmm, seed)
        assert len(ref_hits) > 10
        assert ref_hits == product_tuples(our_hits)


def test_state_checks_abort_and_errors_through_the_reference_wrapper(tmp_path):
    """host/Havac.cpp:80-102,190-192 and host/test/RefernceComparisonTest/ReferenceComparisonTest.cpp:83-88 (the `abort`
    argument of the reference's on-FPGA test): the exception kinds and messages are the reference's own code's, the
    states come back through integration/HavacHwClient.hpp from the device layer."""
    ref = binding.ReferenceHavac(0, 0.02)
    with pytest.raises(binding.RefHostError, match="run object was not initialized") as e:
        ref.currentHardwareState()                                       # HavacHwClient.cpp:163-170 via the shim
    assert e.value.code == -2
    with pytest.raises(binding.RefHostError, match="Phmm was not loaded to device") as e:
        ref.runHardwareClientAsync()                                     # host/Havac.cpp:86-88
    assert e.value.code == -2
    with pytest.raises(binding.RefHostError, match="Could not open fasta file for reading") as e:
        ref.loadSequence(str(tmp_path / "missing.fa"))                   # host/Havac.cpp:62-64
    assert e.value.code == -3

    # a long run: one model of 120,000 rows against 100 Mbp is about a quarter of a second of GPU work
    _, cons = synth.dfam_like_model(120_000, 1)
    hmm = str(tmp_path / "long.hmm")
    synth.write_hmm(hmm, [dict(name="long", acc="RF1", emissions=synth.emissions_from_consensus(cons, 2), maxl=200_000,
                               mu=-9.0, lam=0.7)])
    fa = str(tmp_path / "long.fa")
    rng = np.random.default_rng(3)
    with open(fa, "w") as f:
        f.write(">chr\n")
        letters = np.frombuffer(b"ACGT", np.uint8)
        for _ in range(100):
            f.write(letters[rng.integers(0, 4, size=1_000_000)].tobytes().decode() + "\n")
    ref.loadPhmm(hmm)
    with pytest.raises(binding.RefHostError, match="Sequence was not loaded to device"):
        ref.runHardwareClient()                                          # host/Havac.cpp:89-91
    ref.loadSequence(fa)
    ref.runHardwareClientAsync()
    assert ref.currentHardwareState() == 3                               # RUNNING
    t0 = time.time()
    ref.abortHardwareClient()
    assert time.time() - t0 < 5.0
    assert ref.currentHardwareState() == 6                               # ABORT
    with pytest.raises(binding.RefHostError) as e:                       # a stopped sweep has no hit list
        ref.getHitsFromFinishedRun()
    assert e.value.code == -2
    # the same object runs to completion afterwards
    ref.runHardwareClient()
    assert ref.currentHardwareState() == 4
    assert isinstance(ref.getHitsFromFinishedRun(), list)
    ref.close()
