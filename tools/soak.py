"""Randomised soak of the HIP path against the CPU checker: many small problems of random shape and score
distribution (and random separator masks), compared element for element.  A one-off confidence run, not part
of the test suite:   python tools/soak.py [cases] [seed] [big]
With `big`, problems are 50-2500 segments x up to 25000 rows of Dfam-like models (up to ~6e11 cells each) and the
checker is the oracle's vectorised whole-matrix route."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from havac_amd import synth  # noqa: E402
from havac_amd.hw_client import HavacHwClient  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
big = len(sys.argv) > 3 and sys.argv[3] == "big"
inflight = len(sys.argv) > 4 and sys.argv[4] == "inflight"
rng = np.random.default_rng(seed)
c = HavacHwClient()
if inflight:
    c.setPipelineDepth(2)
bad = 0
t0 = time.time()
cells = hits_total = 0


def make_case(case):
    nseg = int(rng.integers(1, 6))
    n = nseg * synth.SEGMENT
    nrows = int(rng.choice([1, 2, 3, 31, 32, 33, 64, 100, 255, 256, 257, 1000, 2048, 2049, 4100, int(rng.integers(1, 3000))]))
    kind = int(rng.integers(0, 6))
    if big:
        nseg = int(rng.integers(50, 2500))
        n = nseg * synth.SEGMENT
        nrows = int(rng.choice([32, 64, 100, 200, 256, 257, 1024, 2047, 2048, 2049, 5000, 25000, int(rng.integers(1, 300)), int(rng.integers(1, 12000))]))
        kind = int(rng.choice([1, 5]))
    if kind == 0:
        model = rng.integers(-128, 128, size=(nrows, 4)).astype(np.int8)
    elif kind == 1:
        model, cons = synth.dfam_like_model(nrows, int(rng.integers(1 << 30)))
    elif kind == 2:
        model = rng.choice(np.array([-128, 127], np.int8), size=(nrows, 4))
    elif kind == 3:
        model = rng.integers(-20, 60, size=(nrows, 4)).astype(np.int8)
    elif kind == 4:
        model = np.full((nrows, 4), -128, np.int8)
        model[rng.integers(0, nrows, size=max(1, nrows // 3))] = 127
    else:
        model, cons = synth.dfam_like_model(nrows, int(rng.integers(1 << 30)))
        model[rng.integers(0, nrows, size=max(1, nrows // 20)), rng.integers(0, 4)] = -128
    sym = rng.integers(0, 4, size=n, dtype=np.uint8)
    if kind in (1, 5):
        synth.plant_homologs(sym, cons, n, every=int(rng.integers(500, 20000)), length=min(nrows, int(rng.integers(20, 600))),
                             sub=float(rng.uniform(0, 0.3)), seed=int(rng.integers(1 << 30)))
    use_mask = rng.random() < 0.3
    # how the launch hands out its tiles (havac_dev_set_tuning): the library's own rule half of the time, else a random one --
    # partitions or none, how many of the last tiles are cut, finest block, taper; or uniform row blocks for every tile
    tuning = [-1] * 9
    # which kernel (round 4): the library's choice, or forced -- short models (up to 256 rows) then run the standard kernel (0; with
    # tiles_per_item: groups of g tiles, negative: the last round of wave slots single), or the resident-table kernel (1; with
    # tiles_per_item = g >= 1: runs of g tiles throughout instead of tapering rounds)
    tuning[8] = int(rng.choice([-1, -1, 0, 1]))
    if nrows <= 256 and rng.random() < 0.5:
        tuning[1] = int(rng.choice([1, 2, 3, 5, 8, 13, -2, -4, -7]))
    pick = rng.random()
    if pick < 0.35:
        tuning[4:8] = [int(rng.choice([0, 1, 2, 3])), int(rng.choice([0, 1, 2, 6, 64])), int(rng.choice([1024, 2048, 4096])), int(rng.choice([2, 3, 4, 16]))]
    elif pick < 0.5:
        tuning[0] = int(rng.choice([1024, 2048, 4096, 8192]))
        tuning[4] = int(rng.choice([0, 3]))
    pieces, mask = [(0, n)], None
    if use_mask:
        seps = sorted(set((rng.integers(0, n // 2, size=int(rng.integers(1, 30))) * 2).tolist()))
        mask = np.zeros(n // 16, np.uint8)
        for s in seps:
            mask[s // 16] |= 1 << ((s // 2) % 8)
        pieces, start = [], 0
        for s in seps + [n]:
            if s > start:
                pieces.append((start, s))
            start = s + 2
    want = []
    for a, b in pieces:
        if big:
            h = O.ssv_fast(sym[a:b], model, nthreads=16, cap=1 << 22)
        else:
            h = O.ssv_mt(sym[a:b], model) if (b - a) * nrows > 5e7 else O.ssv(sym[a:b], model)
        r, cc = O.unpack_hits(h)
        want.append(O.pack_hits(r, cc + np.uint64(a)))
    want = O.device_order(np.concatenate(want)) if want else np.zeros(0, np.uint64)
    return dict(case=case, nseg=nseg, n=n, nrows=nrows, kind=kind, sym=sym, model=model, mask=mask, tuning=tuning, want=want)


def start_run(k):
    c.writeSequence(synth.pack_2bit(k["sym"]))
    if k["mask"] is not None:
        c.writeSeparatorMask(k["mask"])
    c.writePhmm(k["model"])
    c.invokeHavacSsvAsync()


def fetch_and_compare(k):
    global bad, cells, hits_total
    state = c.waitForHavacSsvAsync()
    got = c.getHitList()
    if inflight:
        c.retire()
    ok = state == 4 and np.array_equal(got, k["want"])
    cells += k["n"] * k["nrows"]
    hits_total += k["want"].size
    if not ok:
        bad += 1
        print(f"MISMATCH case {k['case']}: nseg={k['nseg']} nrows={k['nrows']} kind={k['kind']} mask={k['mask'] is not None} tuning={k['tuning']} "
              f"got {got.size} want {k['want'].size}", flush=True)
    if k["case"] % (5 if big else 25) == (4 if big else 24):
        print(f"{k['case'] + 1} cases, {bad} mismatches, {cells:.3g} cells, {hits_total} hits, {time.time() - t0:.0f} s", flush=True)


case = 0
while case < cases:
    group = [make_case(case + i) for i in range(2 if inflight and case + 1 < cases else 1)]
    c.setTuning(*group[0]["tuning"])
    c.setHitCapacity(max(1 << 16, max(k["want"].size for k in group) + 8))
    for k in group:
        start_run(k)          # (two runs open: the second's inputs are written while the first is in flight)
    for k in group:
        fetch_and_compare(k)
    case += len(group)
c.close()
print(f"SOAK {'FAILED' if bad else 'OK'}: {cases} cases, seed {seed}{', two runs in flight' if inflight else ''}, {bad} mismatches, {cells:.3g} cells, {hits_total} hits compared element for element")
sys.exit(1 if bad else 0)
