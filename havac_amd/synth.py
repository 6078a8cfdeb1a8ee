"""Synthetic inputs for tests and bench.

Everything is generated from documented numpy seeds so that the GPU path and
the CPU checker are always fed the very same arrays.  Nothing here touches a
GPU or the oracle.

* sequence: iid uniform {A,C,G,T} (symbols 0..3), optionally with planted
  homologs of a model's consensus so that hit paths are exercised;
* Dfam-like int8 model: per row one consensus base with a score in [+20,+40],
  the other three in [-60,-30] (what PhmmReprojection yields for typical
  scale ~ 14);
* model-length draws for the 1000-model collection: log-uniform in [50,2000].
"""
from __future__ import annotations

import numpy as np

SEGMENT = 12288  # device/PublicDefines.h:18-22 (NUM_CELL_PROCESSORS)

SEED_SEQUENCE = 1001
SEED_MODEL = 2001
SEED_LENGTHS = 3001
SEED_PLANT = 4001


def padded_length(n: int) -> int:
    """Round up to whole 12288-column segments (SequencePreprocessor.cpp:15-17)."""
    return (n + SEGMENT - 1) // SEGMENT * SEGMENT


def random_symbols(n: int, seed: int = SEED_SEQUENCE, pad: bool = True) -> np.ndarray:
    """n iid symbols; padded with symbol 0 ('A', SequencePreprocessor.cpp:41) to a segment multiple."""
    rng = np.random.default_rng(seed)
    total = padded_length(n) if pad else n
    out = np.zeros(total, dtype=np.uint8)
    out[:n] = rng.integers(0, 4, size=n, dtype=np.uint8)
    return out


def random_packed(nsymbols: int, seed: int = SEED_SEQUENCE) -> np.ndarray:
    """2-bit packed iid sequence, generated directly as bytes (for 10^8..10^9 symbols).

    nsymbols must be a multiple of 12288; every byte carries four uniform symbols."""
    assert nsymbols % SEGMENT == 0
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=nsymbols // 4, dtype=np.uint8)


def pack_2bit(symbols: np.ndarray) -> np.ndarray:
    """Symbol i -> byte i/4, bits (i%4)*2 (SequencePreprocessor.cpp:46-57)."""
    s = np.ascontiguousarray(symbols, dtype=np.uint8)
    assert s.size % 4 == 0
    q = s.reshape(-1, 4)
    return (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)


def unpack_2bit(packed: np.ndarray) -> np.ndarray:
    p = np.ascontiguousarray(packed, dtype=np.uint8)
    out = np.empty((p.size, 4), dtype=np.uint8)
    for k in range(4):
        out[:, k] = (p >> (2 * k)) & 3
    return out.reshape(-1)


def dfam_like_model(nrows: int, seed: int = SEED_MODEL):
    """-> (int8 [nrows,4] scores, uint8 [nrows] consensus symbols)."""
    rng = np.random.default_rng(seed)
    consensus = rng.integers(0, 4, size=nrows, dtype=np.uint8)
    scores = rng.integers(-60, -29, size=(nrows, 4)).astype(np.int8)
    scores[np.arange(nrows), consensus] = rng.integers(20, 41, size=nrows).astype(np.int8)
    return scores, consensus


def model_lengths(count: int = 1000, lo: int = 50, hi: int = 2000, seed: int = SEED_LENGTHS) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return np.exp(rng.uniform(np.log(lo), np.log(hi), size=count)).astype(np.int64)


def model_collection(lengths, seed: int = SEED_MODEL):
    """Concatenate per-model tables back to back, no separator (PhmmPreprocessor.cpp:9-31)."""
    parts, cons = [], []
    for k, L in enumerate(lengths):
        s, c = dfam_like_model(int(L), seed + k)
        parts.append(s)
        cons.append(c)
    return np.concatenate(parts), np.concatenate(cons)


def plant_homologs(symbols: np.ndarray, consensus: np.ndarray, nreal: int, every: int = 1_000_000,
                   length: int = 300, sub: float = 0.15, seed: int = SEED_PLANT) -> int:
    """Overwrite stretches of `symbols` with mutated copies of the consensus; returns how many."""
    rng = np.random.default_rng(seed)
    length = min(length, consensus.size)
    planted = 0
    pos = every // 2
    while pos + length <= nreal:
        start = int(rng.integers(0, consensus.size - length + 1))
        piece = consensus[start:start + length].copy()
        mut = rng.random(length) < sub
        piece[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
        symbols[pos:pos + length] = piece
        planted += 1
        pos += every
    return planted


# ---- synthetic FILES for the file-level API (Havac::loadSequence / loadPhmm) ---------------------

def write_fasta(path: str, records, width: int = 60) -> None:
    """records: iterable of (header, symbols-or-string).  Symbols 0..3 are written as ACGT."""
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "w") as f:
        for header, seq in records:
            if not isinstance(seq, str):
                seq = letters[np.asarray(seq, dtype=np.uint8)].tobytes().decode()
            f.write(f">{header}\n")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + "\n")


def emissions_from_consensus(consensus: np.ndarray, seed: int, lo: float = 0.80, hi: float = 0.97) -> np.ndarray:
    """[L,4] match-emission file values (-ln p) of a Dfam-like model with the given consensus."""
    rng = np.random.default_rng(seed)
    L = consensus.size
    pc = rng.uniform(lo, hi, size=L)
    rest = rng.dirichlet(np.ones(3) * 4, size=L) * (1 - pc)[:, None]
    p = np.empty((L, 4))
    for k in range(L):
        others = [a for a in range(4) if a != consensus[k]]
        p[k, consensus[k]] = pc[k]
        p[k, others] = rest[k]
    return -np.log(p)


def write_hmm(path: str, models, append: bool = False) -> None:
    """models: iterable of dicts {name, acc, emissions [L,4] (-ln p), maxl, mu, lam}.  HMMER3/f ASCII."""
    with open(path, "a" if append else "w") as f:
        for m in models:
            em = np.asarray(m["emissions"], dtype=np.float64)
            L = em.shape[0]
            f.write("HMMER3/f [3.1b2 | February 2015]\n")
            f.write(f"NAME  {m['name']}\n")
            if m.get("acc"):
                f.write(f"ACC   {m['acc']}\n")
            f.write(f"LENG  {L}\n")
            f.write(f"MAXL  {m['maxl']}\n")
            f.write("ALPH  DNA\nRF    no\nMM    no\nCONS  yes\nCS    no\nMAP   yes\n")
            f.write("NSEQ  10\nEFFN  1.500000\nCKSUM 12345\n")
            f.write(f"STATS LOCAL MSV      {m['mu']:9.4f} {m['lam']:8.5f}\n")
            f.write(f"STATS LOCAL VITERBI  {m['mu'] - 0.8:9.4f} {m['lam']:8.5f}\n")
            f.write(f"STATS LOCAL FORWARD  {m['mu'] + 5.0:9.4f} {m['lam']:8.5f}\n")
            f.write("HMM          A        C        G        T   \n")
            f.write("            m->m     m->i     m->d     i->m     i->i     d->m     d->d\n")
            f.write("  COMPO   1.38629  1.38629  1.38629  1.38629\n")
            f.write("          1.38629  1.38629  1.38629  1.38629\n")
            f.write("          0.01005  5.29832  5.29832  0.61958  0.77255  0.00000        *\n")
            for k in range(L):
                vals = "  ".join("      *" if not np.isfinite(v) else f"{v:7.5f}" for v in em[k])
                cons = "acgt"[int(np.argmin(em[k]))]
                f.write(f"{k + 1:7d}   {vals} {k + 1:6d} {cons} - - -\n")
                f.write("          1.38629  1.38629  1.38629  1.38629\n")
                last = "        *" if k == L - 1 else "  0.95510"
                f.write(f"          0.01005  5.29832  5.29832  0.61958  0.77255  0.48576{last}\n")
            f.write("//\n")
