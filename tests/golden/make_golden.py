"""Generates tests/golden/*.npz from the REFERENCE's softSsvThreshold256.

Run in the build container only (needs oracle/_ref, i.e. /root/reference):

    make -C oracle all _ref && python tests/golden/make_golden.py

Each fixture is data only: the 2-bit packed sequence handed to the device, the
flattened int8 model, and the reference's hits as packed 64-bit records
(device/HitReporting.cpp:421-430) in device order.  The reference itself ships
no golden vectors, so these pin both the CPU
restatement (oracle/ssv_oracle.c) and the HIP path.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O          # noqa: E402
from havac_amd import synth               # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEG = synth.SEGMENT


def save(name, symbols, model, note):
    symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
    assert symbols.size % SEG == 0
    model = np.ascontiguousarray(model, dtype=np.int8).reshape(-1, 4)
    hits = O.ssv_reference(symbols, model)
    ours = O.ssv(symbols, model)
    assert np.array_equal(hits, ours), name
    np.savez_compressed(os.path.join(OUT, name + ".npz"), packed=synth.pack_2bit(symbols),
                        model=model, hits=hits, note=np.array(note))
    print(f"{name}: rows={model.shape[0]} cols={symbols.size} hits={hits.size}  {note}")


def planted(symbols, consensus, at, start, length, sub, rng):
    piece = consensus[start:start + length].copy()
    mut = rng.random(length) < sub
    piece[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
    symbols[at:at + length] = piece


def main():
    if not O.ref_available():
        sys.exit("oracle/_ref is not built; fixtures can only be made where /root/reference exists")
    rng = np.random.default_rng(20261003)

    # G1: config C1 -- one model L=100 x 10 kbp (padded to one segment)
    model, cons = synth.dfam_like_model(100, 2001)
    sym = synth.random_symbols(10_000, 1001)
    planted(sym, cons, 4000, 0, 100, 0.1, rng)
    save("g1_c1_L100_10kbp", sym, model, "C1: L=100 x 10 kbp, one planted homolog")

    # G2: L=1024 x 3 segments, homolog crossing the segment boundary at column 12288
    model, cons = synth.dfam_like_model(1024, 2002)
    sym = synth.random_symbols(3 * SEG, 1002)
    planted(sym, cons, SEG - 150, 300, 300, 0.15, rng)
    planted(sym, cons, 2 * SEG - 20, 0, 500, 0.1, rng)
    save("g2_L1024_3seg_cross", sym, model, "L=1024 x 36864, homologs crossing columns 12288 and 24576")

    # G3: three concatenated models (50, 200, 1000), diagonal crossing a model boundary
    model, cons = synth.model_collection([50, 200, 1000], 2003)
    sym = synth.random_symbols(2 * SEG, 1003)
    planted(sym, cons, 3000, 0, 250, 0.05, rng)        # spans model 0 -> model 1
    planted(sym, cons, 15000, 200, 400, 0.1, rng)      # spans model 1 -> model 2
    save("g3_multi_50_200_1000", sym, model, "3 concatenated models, diagonals run across model boundaries")

    # G4: long model L=20000 (config C5 shape) x 3 segments
    model, cons = synth.dfam_like_model(20000, 2004)
    sym = synth.random_symbols(3 * SEG, 1004)
    planted(sym, cons, 100, 0, 2000, 0.15, rng)
    planted(sym, cons, 20000, 15000, 5000, 0.2, rng)
    save("g4_long_L20000", sym, model, "L=20000 x 36864, long planted diagonals")

    # G5: adversarial tables
    sym = synth.random_symbols(SEG, 1005)
    save("g5a_all_plus127", sym, np.full((7, 4), 127, np.int8), "every score +127: a hit every third row on every diagonal")
    save("g5b_all_minus128", sym, np.full((64, 4), -128, np.int8), "every score -128: no hits")
    m = np.zeros((5, 4), np.int8); m[0] = 127; m[1] = 127; m[2] = [2, 1, 0, -1]; m[3] = 127; m[4] = [1, 2, 127, -128]
    save("g5c_256_vs_255", sym, m, "254+2 hits, 254+1 stays at 255, then 255+127 hits")
    save("g5d_random_full_range", synth.random_symbols(SEG, 1006),
         rng.integers(-128, 128, size=(300, 4)).astype(np.int8), "uniform int8 scores: many resets and hits")

    # G6: edges -- hits forced at column 0/row 0 are impossible (one add of <=127), so force the
    # earliest possible ones (row 2) on the first columns, the last column and the last row
    m = np.full((3, 4), 127, np.int8)
    sym = np.zeros(SEG, np.uint8)
    save("g6a_edges_3rows", sym, m, "3 rows of +127 on all-A: hits on row 2 at every column >= 2, incl. the last")
    m = np.full((40, 4), -100, np.int8); m[-3:] = 127
    save("g6b_last_rows", synth.random_symbols(SEG, 1007), m, "only the last three rows score: hits on the final row")


if __name__ == "__main__":
    main()
