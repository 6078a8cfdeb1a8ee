"""The reference's API surface on the GPU: HavacHwClient behaviours (host/HavacHwClient.cpp) and the
file-level `Havac` class (host/Havac.cpp) end to end, checked against the CPU checker."""
import ctypes as C
import time

import numpy as np
import pytest

from havac_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def client():
    from havac_amd.hw_client import HavacHwClient
    c = HavacHwClient(deviceIndex=0)
    yield c
    c.close()


def small_inputs(nrows=200, nseg=2, seed=5):
    model, cons = synth.dfam_like_model(nrows, seed)
    sym = synth.random_symbols(nseg * synth.SEGMENT, seed + 1)
    synth.plant_homologs(sym, cons, sym.size, every=5000, length=min(150, nrows))
    return sym, model


# ---- HavacHwClient argument checks, with the reference's exception types --------------------------

def test_write_sequence_must_be_whole_segments(client):
    from havac_amd.hw_client import LengthError
    with pytest.raises(LengthError, match="multiple of the sequence segment length 12288"):
        client.writeSequence(np.zeros(100, np.uint8))            # host/HavacHwClient.cpp:83-90


def test_write_phmm_must_be_multiple_of_four(client):
    from havac_amd.hw_client import LengthError
    with pytest.raises(LengthError, match="not divisible by 4"):
        client.writePhmm(np.zeros(7, np.int8))                   # host/HavacHwClient.cpp:113-119


def test_run_needs_both_inputs(client):
    from havac_amd.hw_client import LengthError
    with pytest.raises(LengthError, match="sequence length in segments cannot be 0"):
        client.invokeHavacSsvAsync()                             # host/HavacHwClient.cpp:142-144
    client.writeSequence(np.zeros(synth.SEGMENT // 4, np.uint8))
    with pytest.raises(LengthError, match="phmm length in vectors cannot be 0"):
        client.invokeHavacSsvAsync()                             # :145-147


def test_state_before_any_run_is_a_logic_error(client):
    from havac_amd.hw_client import LogicError
    with pytest.raises(LogicError, match="run object was not initialized"):
        client.getHwState()                                      # host/HavacHwClient.cpp:163-170


def test_async_run_states_and_rerun(client, oracle):
    sym, model = small_inputs()
    client.writeSequence(synth.pack_2bit(sym))
    client.writePhmm(model)
    client.invokeHavacSsvAsync()
    assert client.getHwState() in (3, 4)                         # RUNNING or COMPLETED
    assert client.waitForHavacSsvAsync() == 4
    assert client.getHwState() == 4
    want = oracle.ssv(sym, model)
    first = client.getHitList()
    assert np.array_equal(first, want)
    assert client.getNumHits() == want.size
    ssv_ms, total_ms = client.lastRunMs()
    assert 0 < ssv_ms <= total_ms
    # a second run on the same handle, new model, gives the new answer (buffers are reused)
    model2, _ = synth.dfam_like_model(77, 99)
    client.writePhmm(model2)
    client.invokeHavacSsvAsync()
    client.waitForHavacSsvAsync()
    assert np.array_equal(client.getHitList(), oracle.ssv(sym, model2))


def test_hit_buffer_overflow_is_reported(client, oracle):
    from havac_amd.hw_client import HitOverflowError
    sym = np.zeros(synth.SEGMENT, np.uint8)
    model = np.full((9, 4), 127, np.int8)                         # a hit on every third row of every diagonal
    want = oracle.ssv(sym, model)
    client.setHitCapacity(1000)
    client.writeSequence(synth.pack_2bit(sym))
    client.writePhmm(model)
    client.invokeHavacSsvAsync()
    assert client.waitForHavacSsvAsync() == 5                     # ERROR
    with pytest.raises(HitOverflowError):
        client.getHitList()
    client.setHitCapacity(want.size)                              # exactly enough
    client.invokeHavacSsvAsync()
    assert client.waitForHavacSsvAsync() == 4
    assert np.array_equal(client.getHitList(), want)


def test_dense_hits_every_cell_region(client, oracle):
    """Every row +127 on all-A: a third of all cells hit; exercises the staging/flush path hard."""
    sym = synth.random_symbols(2 * synth.SEGMENT, 3)
    model = np.full((200, 4), 127, np.int8)
    want = oracle.ssv_mt(sym, model)
    client.setHitCapacity(want.size + 10)
    client.writeSequence(synth.pack_2bit(sym))
    client.writePhmm(model)
    client.invokeHavacSsvAsync()
    client.waitForHavacSsvAsync()
    got = client.getHitList()
    assert got.size == want.size and np.array_equal(got, want)


def test_abort_and_timeout(client):
    """A long run (about half a second of GPU work) is timed out on, then aborted (HavacHwClient.cpp:153-161)."""
    model, _ = synth.dfam_like_model(120_000, 1)                  # 120k rows x 100 Mbp ~ 1.2e13 cells
    packed = synth.random_packed(8139 * synth.SEGMENT, 2)
    client.writeSequence(packed)
    client.writePhmm(model)
    client.invokeHavacSsvAsync()
    t0 = time.time()
    assert client.waitForHavacSsvAsync(timeout_ms=20) == 8        # TIMEOUT
    assert client.getHwState() == 3                               # still RUNNING
    assert client.abort() == 6                                    # ABORT
    assert time.time() - t0 < 5.0
    assert client.getHwState() == 6
    with pytest.raises(RuntimeError):                             # an aborted sweep has no hit list
        client.getHitList()
    # the handle is usable again afterwards
    sym, small = small_inputs()
    client.writeSequence(synth.pack_2bit(sym))
    client.writePhmm(small)
    client.invokeHavacSsvAsync()
    assert client.waitForHavacSsvAsync() == 4


def test_runs_in_flight_behind_the_handle(oracle):
    """havac_dev_set_pipeline_depth (round 5; no counterpart in the reference, whose client runs one pass at a time,
    host/HavacHwClient.cpp:141-157): three runs with three different models, two of them open at once -- the model is rewritten
    while the run before is still on the device --, results in submission order, each list equal to the checker's for ITS model;
    then the one-at-a-time behaviour again at depth 1."""
    from havac_amd.hw_client import HavacHwClient, LogicError
    sym, model_a = small_inputs(nrows=300, nseg=40, seed=11)
    model_b, _ = synth.dfam_like_model(77, 99)
    model_c, cons_c = synth.dfam_like_model(1500, 5)
    want = [oracle.ssv_fast(sym, m) for m in (model_a, model_b, model_c)]
    assert want[0].size > 50 and not np.array_equal(want[0], want[1])
    c = HavacHwClient()
    c.setHitCapacity(1 << 20)
    c.setPipelineDepth(2)
    c.writeSequence(synth.pack_2bit(sym))
    c.writePhmm(model_a)
    c.invokeHavacSsvAsync()
    c.writePhmm(model_b)                        # while run A is on the device: waits until A has read its model, not until A is done
    c.invokeHavacSsvAsync()
    assert c.openRuns() == 2
    with pytest.raises(LogicError, match="during a run"):
        c.setPipelineDepth(3)
    assert c.waitForHavacSsvAsync() == 4
    assert np.array_equal(c.getHitList(), want[0])              # the OLDEST run: model A
    assert np.array_equal(c.getHitList(), want[0])              # ... as often as asked
    c.writePhmm(model_c)
    c.invokeHavacSsvAsync()                                     # both slots open, the oldest finished: it is closed to make room
    assert c.openRuns() == 2
    assert c.waitForHavacSsvAsync() == 4
    assert np.array_equal(c.getHitList(), want[1])              # model B
    ssv_ms, total_ms = c.lastRunMs()
    assert 0 < ssv_ms <= total_ms
    c.retire()
    assert c.openRuns() == 1 and c.getHwState() in (3, 4)
    assert np.array_equal(c.getHitList(), want[2])              # model C (getHitList waits)
    c.retire()
    assert c.openRuns() == 0
    with pytest.raises(LogicError):
        c.retire()
    # abort with two runs in flight stops both
    long_model, _ = synth.dfam_like_model(60_000, 1)
    c.setHitCapacity(1 << 27)                                   # (a stopped sweep may still have queued more than the small buffer above holds)
    c.writeSequence(synth.random_packed(2000 * synth.SEGMENT, 2))
    c.writePhmm(long_model)
    c.invokeHavacSsvAsync()
    c.invokeHavacSsvAsync()
    assert c.abort() == 6
    c.retire()
    assert c.waitForHavacSsvAsync() == 6
    c.retire()
    # back to one run at a time: the reference's behaviour, a finished run is closed by the next one
    c.setPipelineDepth(1)
    c.writeSequence(synth.pack_2bit(sym))
    c.writePhmm(model_b)
    c.invokeHavacSsvAsync()
    assert c.waitForHavacSsvAsync() == 4 and np.array_equal(c.getHitList(), want[1])
    c.writePhmm(model_a)
    c.invokeHavacSsvAsync()
    assert np.array_equal(c.getHitList(), want[0])
    c.close()


def test_an_overflowing_run_between_two_good_ones(oracle):
    """Three runs, two open at once; the middle one finds more records than the hit buffer holds: it alone ends in ERROR (5) with
    HitOverflowError on its list, the runs before and after it are unaffected (every run has its own slot: context, hit buffer,
    report) -- the reference's client has one buffer and one run, host/HavacHwClient.cpp:141-202."""
    from havac_amd.hw_client import HavacHwClient, HitOverflowError
    sym, model_a = small_inputs(nrows=300, nseg=12, seed=5)
    dense = np.full((40, 4), 127, np.int8)                         # a hit on every third row of every diagonal: far more than the buffer
    model_c, _ = synth.dfam_like_model(500, 17)
    want_a, want_c = oracle.ssv_fast(sym, model_a), oracle.ssv_fast(sym, model_c)
    capacity = max(want_a.size, want_c.size) + 64
    assert oracle.ssv_fast(sym, dense).size > 4 * capacity
    c = HavacHwClient()
    c.setHitCapacity(capacity)
    c.setPipelineDepth(2)
    c.writeSequence(synth.pack_2bit(sym))
    c.writePhmm(model_a)
    c.invokeHavacSsvAsync()
    c.writePhmm(dense)
    c.invokeHavacSsvAsync()
    assert c.waitForHavacSsvAsync() == 4 and np.array_equal(c.getHitList(), want_a)
    c.retire()
    c.writePhmm(model_c)
    c.invokeHavacSsvAsync()                                         # beside the overflowing run
    assert c.waitForHavacSsvAsync() == 5                            # the oldest open run: the dense one -- ERROR
    with pytest.raises(HitOverflowError):
        c.getHitList()
    c.retire()
    assert c.waitForHavacSsvAsync() == 4 and np.array_equal(c.getHitList(), want_c)
    c.retire()
    assert c.openRuns() == 0
    c.close()


def test_runs_in_flight_over_several_gpus(oracle):
    """the same with four device parts behind the handle (one GPU named four times): every part keeps two passes in flight"""
    from havac_amd.hw_client import HavacHwClient
    sym, model_a = small_inputs(nrows=400, nseg=16, seed=21)
    model_b, _ = synth.dfam_like_model(130, 8)
    want = [oracle.ssv_fast(sym, m) for m in (model_a, model_b)]
    c = HavacHwClient(deviceIndices=[0, 0, 0, 0])
    c.setHitCapacity(1 << 18)
    c.setPipelineDepth(2)
    c.writeSequence(synth.pack_2bit(sym))
    for k in range(6):
        c.writePhmm(model_a if k % 2 == 0 else model_b)
        c.invokeHavacSsvAsync()
        if c.openRuns() == 2:
            assert np.array_equal(c.getHitList(), want[(k - 1) % 2])
            c.retire()
    assert np.array_equal(c.getHitList(), want[1])
    c.close()


# ---- the file-level Havac class --------------------------------------------------------------------

def write_inputs(tmp_path, lengths, record_lengths, seed=0):
    rng = np.random.default_rng(seed)
    models, all_cons = [], []
    for k, L in enumerate(lengths):
        _, cons = synth.dfam_like_model(L, 70 + k + seed)
        all_cons.append(cons)
        models.append(dict(name=f"fam{k}", acc=f"RF{k:05d}", emissions=synth.emissions_from_consensus(cons, 80 + k),
                           maxl=3 * L + 50, mu=-9.2 + 0.1 * k, lam=0.71))
    cons = np.concatenate(all_cons)
    records = []
    for k, n in enumerate(record_lengths):
        s = rng.integers(0, 4, size=n, dtype=np.uint8)
        synth.plant_homologs(s, cons, n, every=4000, length=min(cons.size, 250), seed=k)
        text = "".join("ACGT"[v] for v in s)
        if k == 1 and n > 50:
            text = text[:20] + "NNNNRYKM" + text[28:]             # ambiguity codes draw from rand()
        records.append((f"seq{k}", text))
    fa, hmm = tmp_path / "in.fa", tmp_path / "in.hmm"
    synth.write_fasta(str(fa), records)
    synth.write_hmm(str(hmm), models)
    return str(fa), str(hmm)


def test_havac_class_end_to_end(tmp_path, oracle):
    from havac_amd import havac
    fa, hmm = write_inputs(tmp_path, [60, 300, 150], [5000, 9000, 30000, 17])
    libc = C.CDLL(None)
    seed = 4242
    packed, nchars, nrec = havac.pack_fasta(fa, seed=seed)        # what loadSequence will send, given the same srand
    table, lens = havac.project_hmm(hmm, 0.02)
    want_raw = oracle.ssv(oracle.unpack_2bit(packed), table)
    assert want_raw.size > 20

    h = havac.Havac(0, 0.02)
    from havac_amd.hw_client import LogicError
    with pytest.raises(LogicError, match="Phmm was not loaded"):
        h.runHardwareClient()                                     # host/Havac.cpp:86-88
    h.loadPhmm(hmm)
    with pytest.raises(LogicError, match="Sequence was not loaded"):
        h.runHardwareClientAsync()                                # :89-91
    libc.srand(seed)
    h.loadSequence(fa)
    h.runHardwareClient()
    assert h.currentHardwareState() == 4
    hits = h.getHitsFromFinishedRun()
    assert np.array_equal(h.rawHits(), want_raw)                  # device order, element for element
    # The resolved hits, worked out here and not by the product's resolver (host/Havac.cpp:104-187 restated in numpy):
    # global column -> (record, position in record) through the records' end positions (each record occupies its
    # residues plus one terminator column; a column at or beyond the last end is padding and the hit is dropped),
    # global row -> (model, position in model) through the prefix sums of the model lengths.
    record_lengths = [5000, 9000, 30000, 17]
    record_ends = np.cumsum([n + 1 for n in record_lengths])        # sequenceEndPosition counts the terminator
    model_starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    rows_w, cols_w = oracle.unpack_hits(want_raw)
    expected = []
    for row, col in zip(rows_w.tolist(), cols_w.tolist()):
        if col >= record_ends[-1]:
            continue                                                 # padding behind the last record
        rec = int(np.searchsorted(record_ends, col, side="right"))
        model = int(np.searchsorted(model_starts, row, side="right")) - 1
        expected.append((col - (int(record_ends[rec - 1]) if rec else 0), rec, row - int(model_starts[model]), model))
    assert [(x.sequencePosition, x.sequenceIndex, x.phmmPosition, x.phmmIndex) for x in hits] == expected
    assert hits == havac.resolve_hits(fa, hmm, want_raw)          # and the free-standing resolver agrees
    # every resolved hit points into a real record and a real model
    ends = np.cumsum(lens)
    for x in hits:
        assert x.sequenceIndex < nrec and x.phmmIndex < len(lens) and x.phmmPosition < lens[x.phmmIndex]
    # padding hits (beyond the last record) are dropped, nothing else
    _, cols = oracle.unpack_hits(want_raw)
    assert len(hits) == int((cols < nchars).sum())
    with pytest.raises(RuntimeError):
        h.loadPhmm(str(tmp_path / "missing.hmm"))
    with pytest.raises(RuntimeError, match="Could not open fasta"):
        h.loadSequence(str(tmp_path / "missing.fa"))
    h.close()


def test_havac_class_with_runs_in_flight(tmp_path, oracle):
    """Havac::setPipelineDepth: the next model file is loaded and its run started before the hits of the run before are
    fetched; every run's hits are resolved against the models IT ran with (the model list has been replaced by then)."""
    from havac_amd import havac
    fa, hmm1 = write_inputs(tmp_path, [60, 300, 150], [5000, 9000, 30000, 17])
    sub = tmp_path / "two"
    sub.mkdir()
    _, hmm2 = write_inputs(sub, [220, 90], [100], seed=5)
    single = []
    for hmm in (hmm1, hmm2):
        h = havac.Havac(0, 0.02)
        h.setDevicePacking(False)
        h.loadPhmm(hmm)
        C.CDLL(None).srand(7)
        h.loadSequence(fa)
        h.runHardwareClient()
        single.append((h.getHitsFromFinishedRun(), h.rawHits()))
        h.close()
    assert len(single[0][0]) > 10 and single[0][0] != single[1][0]
    h = havac.Havac(0, 0.02)
    h.setDevicePacking(False)
    h.setPipelineDepth(2)
    C.CDLL(None).srand(7)
    h.loadSequence(fa)
    h.loadPhmm(hmm1)
    h.runHardwareClientAsync()
    h.loadPhmm(hmm2)                                              # replaces the model list while run 1 is open
    h.runHardwareClientAsync()
    first = h.getHitsFromFinishedRun()
    assert first == single[0][0] and np.array_equal(h.rawHits(), single[0][1])
    second = h.getHitsFromFinishedRun()
    assert second == single[1][0] and np.array_equal(h.rawHits(), single[1][1])
    h.close()


def test_havac_benchmark_repeats_with_runs_in_flight(tmp_path):
    """`havac_benchmark --repeat N --depth D` (files through the Havac class) and `--raw` (the C ABI directly, on one of
    bench.py's synthetic workloads written by tools/dump_workload.py): no Python in the process that runs."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    from havac_amd import havac
    fa, hmm = write_inputs(tmp_path, [100, 250], [200000], seed=3)
    exe = os.path.join(os.path.dirname(havac.HOST_LIB_PATH), "havac_benchmark")
    out = subprocess.run([exe, fa, hmm, "--repeat", "20", "--depth", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "20 more runs, 2 in flight" in out.stdout and "the same count every run: yes" in out.stdout
    prefix = str(tmp_path / "w")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dump_workload.py"), "--rows", "256", "--columns-per-gpu", str(400 * synth.SEGMENT), prefix],
                   check=True, capture_output=True, timeout=300)
    for depth in ("1", "2"):
        out = subprocess.run([exe, "--raw", prefix + ".seq", prefix + ".model", "--repeat", "30", "--depth", depth], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "the same list every run: yes" in out.stdout and f"30 runs, {depth} in flight" in out.stdout and "GCUPS" in out.stdout


def test_the_cxx_drop_in_reaches_the_engines_rate_on_c2(tmp_path):
    """"a C++ test (havac_benchmark --repeat 50) reaching within 1 % of bench.py's value on C2 with no
    Python in the process".  bench.py's C2 workload as files, `havac_benchmark --raw --repeat 100 --depth 3 --no-readback` (the
    handle API: havac_dev_run_async / _wait / _num_hits64 / _retire), and the same passes through bench.py's own engine
    (ShardedSsv = havac_pipe_run) in this process, on the same GPU one after the other.  Asserted at 3 % -- the two are measured
    seconds apart, and boxes wander by a per cent; the profiles (r05j_*) have them within 0.3 %."""
    import os
    import re
    import subprocess
    import sys
    import time
    import torch
    from conftest import ROOT
    from havac_amd import havac
    from havac_amd.dist import ShardedSsv
    sys.path.insert(0, ROOT)
    import bench
    prefix = str(tmp_path / "c2")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dump_workload.py"), "--workload", "c2", prefix], check=True, capture_output=True, timeout=300)
    exe = os.path.join(os.path.dirname(havac.HOST_LIB_PATH), "havac_benchmark")
    out = subprocess.run([exe, "--raw", prefix + ".seq", prefix + ".model", "--repeat", "100", "--depth", "3", "--no-readback"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "the same list every run: yes" in out.stdout and "1012547 hits per run" in out.stdout, out.stdout
    cxx = float(re.search(r"= ([0-9.]+) GCUPS", out.stdout).group(1))
    model, packed, ncols, _, _ = bench.make_inputs("c2", 1, 0, 0)
    dev = torch.device("cuda", 0)
    d_seq, d_phmm = torch.from_numpy(packed).to(dev), torch.from_numpy(model.reshape(-1)).to(dev)
    eng = ShardedSsv(1 << 22, dev, depth=3)
    eng.run_many(20, d_seq, ncols, d_phmm, model.shape[0])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    (_, found), _ = eng.run_many(100, d_seq, ncols, d_phmm, model.shape[0], inputs_ready=True)
    torch.cuda.synchronize(dev)
    engine = ncols * model.shape[0] / ((time.perf_counter() - t0) / 100) / 1e9
    eng.release()
    assert found == 1012547
    assert cxx > 50000 and cxx >= 0.97 * engine, (cxx, engine)


def test_havac_benchmark_binary(tmp_path):
    """The counterpart of benchmark/benchmark.cpp runs and prints the reference's timing lines."""
    import os
    import subprocess
    from havac_amd import havac
    fa, hmm = write_inputs(tmp_path, [100], [20000], seed=3)
    exe = os.path.join(os.path.dirname(havac.HOST_LIB_PATH), "havac_benchmark")
    out = subprocess.run([exe, fa, hmm], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    for line in ("verified hits.", "havac build time", "havac load time", "havac run time", "havac verify time", "total time taken"):
        assert line in out.stdout


def test_limits_of_the_record_format(client):
    """rows < 2^24 (24-bit row field, device/HavacHls.hpp:19); packed sequence < 4 GiB (HavacHwClient.cpp:92-97);
    model bytes < 1 GiB (:121-125).  All refused before anything is uploaded."""
    from havac_amd.hw_client import LengthError
    with pytest.raises(LengthError, match="less than 4GiB"):
        client.writeSequence(np.zeros(-(-(4 << 30) // 3072) * 3072, np.uint8))   # whole segments, just over 4 GiB
    with pytest.raises(LengthError, match="model length must be less than"):
        client.writePhmm(np.zeros(1 << 30, np.int8))
    client.writeSequence(np.zeros(synth.SEGMENT // 4, np.uint8))
    client.writePhmm(np.zeros(((1 << 24) * 4,), np.int8))          # 2^24 rows: one more than the record can name
    with pytest.raises(LengthError, match="below 2\\^24"):
        client.invokeHavacSsvAsync()


def test_one_handle_over_several_gpus(oracle, tmp_path):
    """havac_dev_create_multi: several device parts behind one handle, one column shard each.
    The box has one GPU, so the same device is named several times: same code path, same answer, same order."""
    from havac_amd.hw_client import HavacHwClient, LengthError
    sym, model = small_inputs(nrows=700, nseg=7, seed=31)
    want = oracle.ssv_mt(sym, model)
    for devices in ([0, 0], [0, 0, 0], [0] * 7):
        c = HavacHwClient(deviceIndices=devices)
        c.setHitCapacity(want.size + 8)
        c.writeSequence(synth.pack_2bit(sym))
        c.writePhmm(model)
        c.invokeHavacSsvAsync()
        assert c.waitForHavacSsvAsync() == 4
        assert c.getNumHits() == want.size
        assert np.array_equal(c.getHitList(), want)
        c.close()
    c = HavacHwClient(deviceIndices=[0] * 8)                      # more GPUs than segments
    c.writeSequence(synth.pack_2bit(sym))
    c.writePhmm(model)
    with pytest.raises(LengthError, match="fewer 12288-column segments"):
        c.invokeHavacSsvAsync()
    c.close()
    # and through the C++ Havac class
    from havac_amd import havac
    fa, hmm = write_inputs(tmp_path, [90, 310], [20000, 30000, 9000], seed=17)
    single = havac.Havac(0, 0.02)
    multi = havac.Havac(requiredPValue=0.02, deviceIndices=[0, 0, 0])
    import ctypes as C
    for h in (single, multi):
        h.loadPhmm(hmm)
        C.CDLL(None).srand(5)
        h.loadSequence(fa)
        h.runHardwareClient()
    a, b = single.getHitsFromFinishedRun(), multi.getHitsFromFinishedRun()
    assert len(a) > 20 and a == b
    assert np.array_equal(single.rawHits(), multi.rawHits())
    single.close(); multi.close()


def test_gpus_of_a_handle_hold_column_windows_only(oracle):
    """Several GPUs behind one handle: GPU i keeps only the columns shard i can read (its own + a halo for the tallest model
    the record format allows, 2^24 - 1 rows), not the whole database -- at the reference's 4 GiB limit eight full copies
    would be 32 GiB.  40.5 M columns over four parts: the last three windows start well inside the database.  The hit list
    equals the checker's, with and without a separator bitmap, and the sequence reads back whole."""
    from havac_amd.hw_client import HavacHwClient
    nseg = 3300
    packed = synth.random_packed(nseg * synth.SEGMENT, 77)
    model, cons = synth.dfam_like_model(300, 78)
    sym = synth.unpack_2bit(packed)
    synth.plant_homologs(sym, cons, sym.size, every=50_000, length=250, sub=0.08)
    packed = synth.pack_2bit(sym)
    want = oracle.ssv_fast(sym, model, nthreads=8)
    assert want.size > 2000
    c = HavacHwClient(deviceIndices=[0, 0, 0, 0])
    c.writeSequence(packed)
    c.writePhmm(model)
    c.invokeHavacSsvAsync()
    assert c.waitForHavacSsvAsync() == 4
    assert np.array_equal(c.getHitList(), want)
    assert np.array_equal(c.readSequence(packed.size), packed)
    # separators at a few even columns, among them the first and last pair of every shard
    rng = np.random.default_rng(5)
    quarter = nseg // 4 * synth.SEGMENT
    seps = sorted(set((rng.integers(1, sym.size // 2 - 1, size=40) * 2).tolist() + [0, quarter - 2, quarter, 2 * quarter, sym.size - 2]))
    mask = np.zeros(sym.size // 16, np.uint8)
    for col in seps:
        mask[col // 16] |= 1 << ((col // 2) % 8)
    c.writeSeparatorMask(mask)
    c.invokeHavacSsvAsync()
    assert c.waitForHavacSsvAsync() == 4
    got = c.getHitList()
    assert np.array_equal(c.readSeparatorMask(mask.size), mask)
    c.close()
    single = HavacHwClient()
    single.writeSequence(packed)
    single.writeSeparatorMask(mask)
    single.writePhmm(model)
    single.invokeHavacSsvAsync()
    single.waitForHavacSsvAsync()
    assert np.array_equal(got, single.getHitList()) and got.size != want.size
    single.close()


def test_deferred_start_gives_the_same_hits_and_reports_a_missing_device_late(tmp_path):
    """Havac::DeferredStart (an addition): the constructor returns at once, the HIP start-up runs under loadPhmm /
    loadSequence, which wait for the device only when they have something to send."""
    import ctypes as C
    from havac_amd import havac
    from havac_amd.hw_client import LogicError
    fa, hmm = write_inputs(tmp_path, [60, 300], [5000, 9000, 17], seed=3)
    lists = []
    for deferred in (False, True):
        t0 = time.time()
        h = havac.Havac(0, 0.02, deferredStart=deferred)
        built = time.time() - t0
        if deferred:
            assert built < 0.05
            with pytest.raises(LogicError, match="Phmm was not loaded"):
                h.runHardwareClient()
        h.loadPhmm(hmm)
        C.CDLL(None).srand(9)
        h.loadSequence(fa)
        h.runHardwareClient()
        lists.append((h.getHitsFromFinishedRun(), h.rawHits()))
        h.close()
    assert len(lists[0][0]) > 10 and lists[0][0] == lists[1][0] and np.array_equal(lists[0][1], lists[1][1])
    late = havac.Havac(99, 0.02, deferredStart=True)          # no such device: the constructor cannot know yet
    with pytest.raises(RuntimeError, match="could not open MI355X device 99"):
        late.loadPhmm(hmm)
    late.close()


def test_native_software_testbench(tmp_path):
    """tests/native/software_testbench.cpp: the reference's C-simulation testbench flow
    (device/test/softwareTestbench.cpp:49-306) in C++ against the C ABI, 2 tests x 3 segments as there."""
    import os
    import subprocess
    from conftest import ROOT
    from oracle import pyoracle
    pyoracle.build(with_ref=False)
    exe = str(tmp_path / "software_testbench")
    lib_dir, oracle_dir = os.path.join(ROOT, "havac_amd"), os.path.join(ROOT, "oracle")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "native", "software_testbench.cpp"), "-o", exe,
                    f"-L{lib_dir}", "-lhavac_dev", f"-L{oracle_dir}", "-loracle",
                    f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{oracle_dir}"], check=True)
    out = subprocess.run([exe, "2", "3", "2000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "ALL TESTS PASSED" in out.stdout and out.stdout.count("passed") >= 2


@pytest.mark.parametrize("route", ["c_abi", "torch"])
def test_rccl_gather_of_hit_records_single_rank(oracle, route):
    """The collectives of the N > 1 path on the real backend (nccl = RCCL), as far as one GPU allows: a one-rank
    process group, gather_hits on device tensors issued from a side stream (what ShardedSsv does with passes in
    flight), then the barrier and the MAX all-reduce bench.py uses.  Route c_abi (the default on the nccl backend): the
    records go through libhavac_dev.so's own RCCL calls (include/havac_dev.h level 3, havac_gather_*; torch.distributed only
    carries the communicator's id); route torch: through torch.distributed's point-to-point operations."""
    import os
    import torch
    import torch.distributed as dist
    from havac_amd import dist as hdist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    hdist.set_gather_route(route)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29533" if route == "c_abi" else "29534"
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        recs = oracle.pack_hits(np.arange(1000, dtype=np.uint64) % 77, np.arange(1000, dtype=np.uint64) * 13)
        buf = torch.zeros(4096, dtype=torch.int64, device=dev)
        buf[:1000] = torch.from_numpy(recs.view(np.int64)).to(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            merged, counts = hdist.gather_hits(buf, 1000)
        side.synchronize()
        assert counts == [1000]
        assert np.array_equal(merged.cpu().numpy().view(np.uint64), recs)
        assert (len(hdist._c_gathers) == 1) == (route == "c_abi")          # which route ran
        empty, counts = hdist.gather_hits(buf, 0)
        assert counts == [0] and empty.numel() == 0
        # a caller-supplied receive buffer is used when it is large enough
        mine = torch.zeros(2048, dtype=torch.int64, device=dev)
        merged, counts = hdist.gather_hits(buf, 1000, out=mine)
        torch.cuda.synchronize(dev)
        assert merged.data_ptr() == mine.data_ptr() and np.array_equal(mine[:1000].cpu().numpy().view(np.uint64), recs)
        # a rank whose pass failed: every rank raises after the count exchange
        with pytest.raises(hdist.ShardFailure):
            hdist.gather_hits(buf, hdist.FAILED)
        merged, counts = hdist.gather_hits(buf, 7)                          # ... and the communicator is still good
        torch.cuda.synchronize(dev)
        assert counts == [7] and np.array_equal(merged.cpu().numpy().view(np.uint64), recs[:7])
        dist.barrier()
        t = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
    finally:
        hdist.set_gather_route("c_abi")
        hdist.close_c_gathers()
        dist.destroy_process_group()


def test_rccl_gather_through_the_c_entry_points(oracle):
    """havac_gather_* called directly (ctypes, no torch.distributed at all): the id, a one-rank communicator, counts, records
    into a buffer of the caller's, the misuse the header names (records before counts; a receive buffer that is too small is
    refused with HAVAC_E_LENGTH and leaves the communicator unusable)."""
    import torch
    from havac_amd import _lib
    from havac_amd.dist import RcclGather
    from havac_amd.hw_client import LengthError, LogicError
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    assert RcclGather.rccl_version() >= 20000
    uid = RcclGather.unique_id()
    assert len(uid) == 128 and any(uid)
    g = RcclGather(0, 1, uid)
    recs = oracle.pack_hits(np.arange(5000, dtype=np.uint64) % 300, np.arange(5000, dtype=np.uint64) * 7)
    src = torch.from_numpy(recs.view(np.int64).copy()).to(dev)
    dst = torch.zeros(8192, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    with pytest.raises(LogicError):
        g.records(src.data_ptr(), dst.data_ptr(), dst.numel(), stream)
    assert g.counts(5000, stream) == [5000]
    g.records(src.data_ptr(), dst.data_ptr(), dst.numel(), stream)
    torch.cuda.synchronize(dev)
    assert np.array_equal(dst[:5000].cpu().numpy().view(np.uint64), recs) and int(dst[5000:].abs().sum().item()) == 0
    assert g.counts(-1, stream) == [-1]
    with pytest.raises(RuntimeError):
        g.records(src.data_ptr(), dst.data_ptr(), dst.numel(), stream)
    assert g.counts(5000, stream) == [5000]
    with pytest.raises(LengthError):
        g.records(src.data_ptr(), dst.data_ptr(), 100, stream)
    with pytest.raises(LogicError):
        g.counts(1, stream)
    g.close()
    # bad arguments are refused before RCCL is asked
    h = C.c_void_p()
    L = _lib.load()
    assert L.havac_gather_create(3, 2, C.c_char_p(uid), C.byref(h)) == _lib.E_ARGUMENT
    assert L.havac_gather_create(0, 0, C.c_char_p(uid), C.byref(h)) == _lib.E_ARGUMENT


def test_sequence_packed_on_the_gpu(oracle):
    """havac_dev_write_sequence_chars: text in three staging chunks (70 M characters, ragged
    end), ~0.1 % other characters patched in; the buffer read back equals the plainly packed symbols byte for byte,
    and a run on it gives the oracle's records."""
    from havac_amd.hw_client import HavacHwClient
    rng = np.random.default_rng(17)
    n = 70_000_003
    sym = rng.integers(0, 4, size=n, dtype=np.uint8)
    letters = np.frombuffer(b"ACGTacgt", np.uint8)
    chars = letters[sym + 4 * rng.integers(0, 2, size=n, dtype=np.uint8)]
    cols = np.unique(rng.integers(0, n, size=70_000)).astype(np.uint64)
    chars[cols.astype(np.int64)] = np.frombuffer(b"N\0RYKMSWn*", np.uint8)[rng.integers(0, 10, size=cols.size)]
    syms = rng.integers(0, 4, size=cols.size, dtype=np.uint8)
    sym[cols.astype(np.int64)] = syms
    ncolumns = -(-n // synth.SEGMENT) * synth.SEGMENT
    want = np.zeros(ncolumns, np.uint8)
    want[:n] = sym
    c = HavacHwClient()
    c.writeSequenceChars(chars, cols, syms)
    got = c.readSequence(ncolumns // 4)
    assert np.array_equal(got, synth.pack_2bit(want))
    model, cons = synth.dfam_like_model(64, 5)
    c.writePhmm(model)
    c.setHitCapacity(1 << 20)
    c.invokeHavacSsvAsync()
    assert c.waitForHavacSsvAsync() == 4
    assert np.array_equal(c.getHitList(), oracle.ssv_fast(want, model))
    # a short text: one ragged chunk, no patches, then patch-argument errors
    c.writeSequenceChars(b"ACGTACGTAC")
    assert np.array_equal(c.readSequence(3072)[:3], synth.pack_2bit(np.array([0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 0, 0], np.uint8)))
    assert not c.readSequence(3072)[3:].any()
    with pytest.raises(Exception):
        c.writeSequenceChars(b"ACGT", [3, 2], [0, 0])          # not ascending
    with pytest.raises(Exception):
        c.writeSequenceChars(b"ACGT", [20000], [1])            # outside the padded sequence
    c.close()


def test_havac_packs_on_the_gpu_by_default(tmp_path, oracle):
    """Havac::loadSequence with packing on the GPU (the default) and on the host: same raw records for the same
    rand() seed, with ambiguity codes in the file."""
    import ctypes as C
    from havac_amd import havac
    fa, hmm = write_inputs(tmp_path, [90, 200], [30000, 9000, 40], seed=13)
    runs = []
    for on_gpu in (True, False):
        h = havac.Havac(0, 0.02)
        h.setDevicePacking(on_gpu)
        h.loadPhmm(hmm)
        C.CDLL(None).srand(77)
        h.loadSequence(fa)
        h.runHardwareClient()
        h.getHitsFromFinishedRun()
        runs.append(h.rawHits())
        h.close()
    assert runs[0].size > 0 and np.array_equal(runs[0], runs[1])
    table, lens = havac.project_hmm(hmm, 0.02)
    packed, nchars, nrec = havac.pack_fasta(fa, seed=77)
    assert np.array_equal(runs[0], oracle.ssv_mt(oracle.unpack_2bit(packed), table))


def test_more_than_2_to_32_records_behind_one_handle():
    """Four device parts behind one handle report 4.4e9 records between them (C4 on eight GPUs reaches 4.5e9): the
    reference-shaped 32-bit count says so instead of wrapping, the 64-bit entry points give the count and the records.
    Every score is +127, so each diagonal crosses 256 on every third row: the expected list is known in closed form."""
    import ctypes as C
    from havac_amd import _lib
    from havac_amd.hw_client import HavacHwClient
    nseg, nrows = 100, 10_752
    n = nseg * synth.SEGMENT
    c = HavacHwClient(deviceIndices=[0, 0, 0, 0])
    per_part = n // 4 * (nrows // 3)                             # a third of the shard's cells, give or take the left edge
    c.setHitCapacity(per_part + (1 << 25))
    c.writeSequence(np.zeros(n // 4, np.uint8))
    c.writePhmm(np.full((nrows, 4), 127, np.int8))
    c.invokeHavacSsvAsync()
    assert c.waitForHavacSsvAsync() == 4
    # cell (p, s) is the (min(p, s) + 1)-th of its diagonal and hits iff that is a multiple of 3
    s_ = np.arange(nrows, dtype=np.int64)
    want = int(((s_ + 1) // 3 + np.where((s_ + 1) % 3 == 0, nrows - s_ - 1, 0)).sum()) + (n - nrows) * (nrows // 3)
    assert want > (1 << 32)
    L = _lib.load()
    n32 = C.c_uint32(0)
    assert L.havac_dev_num_hits(c._h, C.byref(n32)) == _lib.E_HIT_OVERFLOW
    assert c.getNumHits() == want
    first = np.empty(6, np.uint64)
    assert L.havac_dev_read_hits64(c._h, first.ctypes.data, 6) == 0
    # device order: segment 0, then rows ascending; row 2 is the first with hits, on columns 2, 3, ... of the segment
    rows, cols = first >> np.uint64(40), first & np.uint64(0x3FFF)
    assert rows.tolist() == [2] * 6 and cols.tolist() == [2, 3, 4, 5, 6, 7]
    c.close()


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver reads, the roofline and cpu_baseline objects included
    (small workload here; the default is config C2)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--rows", "256",
                        "--columns-per-gpu", str(400 * synth.SEGMENT)],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "GCUPS" and d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["cells_per_step"] / d["ms_per_step"] / 1e6) < 0.01 * d["value"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    # HBM bytes per launch come from the counters (two rocprofv3 --pmc passes of a child of bench.py), not a constant
    assert d["roofline"]["traffic"] is not None and d["roofline"]["traffic"] > 0, d["roofline"]["traffic_source"]
    assert 0.5 < d["roofline"]["traffic"] / d["roofline"]["hbm"]["algorithmic_bytes_per_launch"] < 4.0
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["cpu_baseline"]["hits_match_gpu"] is True
    assert d["cpu_baseline"]["vectorised_port"]["whole_hit_list_matches_gpu"] is True


@pytest.mark.parametrize("world,rows,segments,gather", [(2, 256, 400, "gloo"), (4, 1024, 150, "gloo"), (4, 1024, 150, "standin"), (3, 256, 200, "standin")])
def test_bench_starts_its_own_ranks(world, rows, segments, gather):
    """`python bench.py --gpus N` with no launcher around it starts N ranks itself.  On
    a one-GPU box all ranks share the card, which RCCL refuses: `--backend gloo`.  gather = "gloo": the records travel through
    torch.distributed over gloo; gather = "standin" (round 5): through libhavac_dev.so's OWN gather (havac_gather_*: the route a
    run on real GPUs takes) bound to tests/native/librccl_standin.so -- the launcher, the sharding, two passes in flight with
    the gather behind the next kernel, the variable-length exchange, the deadline and the per-rank report are then the ones
    an 8-GPU run uses, only librccl itself is not.  Two ranks with a 256-row model (the resident-table kernel on every shard)
    and four with a 1024-row one (the standard kernel, a 1023-column halo on three of them, two boundary stretches checked
    against the CPU)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--rows", str(rows), "--columns-per-gpu", str(segments * synth.SEGMENT), "--backend", "gloo"]
    if gather == "standin":
        from test_gpu_gather_ranks import build_standin
        cmd += ["--gather-library", build_standin()]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["config"]["columns"] == world * segments * synth.SEGMENT
    dd = d["distributed"]
    assert dd["world"] == world and dd["backend"] == "gloo" and len(dd["per_rank"]) == world
    assert ("havac_gather_*" in dd["gather"] and "librccl_standin.so" in dd["gather"]) == (gather == "standin"), dd["gather"]
    assert [p["rank"] for p in dd["per_rank"]] == list(range(world))
    assert dd["per_rank"][0]["halo_cells"] == 0 and all(p["halo_cells"] == (rows - 1) * rows // 2 for p in dd["per_rank"][1:])
    assert d["kernel"]["name"] == ("ssv_resident_kernel" if rows <= 256 else "ssv_diag_kernel")
    assert d["config"]["passes_in_flight"] == 3 and d["config"]["kernel_streams"] == 2
    assert "strictly serial" in d["config"]["api_path"]
    assert sum(p["records"] for p in dd["per_rank"]) == d["config"]["hits_per_step"]
    assert all(p["kernel_ms"] > 0 for p in dd["per_rank"])
    # rank 0 has checked the gathered list before printing: order, counts, columns per rank, the records around the cut
    parity = dd["parity"]
    assert parity["ok"] and parity["counts_add_up"] and parity["device_order_no_duplicates"] and parity["ranks_inside_their_columns"]
    assert len(parity["boundary_stretches"]) == min(2, world - 1) and parity["boundary_stretches"][0]["boundary_column"] == segments * synth.SEGMENT
    assert all(b["equals_cpu_checker"] and b["records"] > 0 for b in parity["boundary_stretches"])
    assert d["clock_warmup_passes"] >= 3


@pytest.mark.parametrize("how", ["exit", "hang"])
def test_a_rank_that_dies_or_hangs_ends_the_job(how):
    """nothing in the N > 1 path had a deadline -- a rank that died or hung left the others inside a
    collective until the driver's limit.  Three ranks on one GPU over the C-ABI gather (stand-in library); in the middle of the
    timed region rank 2 leaves without a word (`--die-at 2:3`, a test hook of bench.py: os._exit) or stops making calls
    (`2:3:hang`).  The job must end non-zero well inside the test's limit; a hung rank is named, with its stage, by the
    watchdogs (`--deadline 8`).  (The deadline of the gather's own host waits: tests/test_gpu_gather_ranks.py.)"""
    import os
    import subprocess
    import sys
    import time
    from test_gpu_gather_ranks import build_standin
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "6", "--warmup", "1", "--rows", "256",
                        "--columns-per-gpu", str(100 * synth.SEGMENT), "--backend", "gloo", "--gather-library", build_standin(),
                        "--deadline", "8", "--die-at", "2:3" + (":hang" if how == "hang" else ""), "--no-pmc"],
                       capture_output=True, text=True, timeout=600, cwd=root, env=env)
    took = time.time() - t0
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]          # no result line from a broken job
    if how == "hang":
        assert "has been in stage 'timed region: pass 3 of 6'" in r.stderr and "giving up, exit code 3" in r.stderr, r.stderr[-3000:]
    assert took < 240, took
