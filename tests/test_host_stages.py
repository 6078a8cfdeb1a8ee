"""Host-only stages of the `Havac` API (no GPU): FASTA reader + 2-bit packer, HMMER3 reader +
int8 projection, hit resolver.  These run on the CPU in the reference too (host/*.cpp)."""
import ctypes as C
import os

import numpy as np
import pytest

from havac_amd import havac, synth


@pytest.fixture(scope="module")
def libc():
    return C.CDLL(None)


def rand_stream(libc, seed, n):
    libc.srand(seed)
    return [libc.rand() for _ in range(n)]


# ---------------------------------------------------------------- FASTA + packing

def test_fasta_records_terminators_and_padding(tmp_path, libc):
    p = tmp_path / "a.fa"
    p.write_text(">one desc\nACGT\nacgt\n\n>two\r\nGG\r\nTT\r\n>three\nA")
    packed, nchars, nrec = havac.pack_fasta(str(p), seed=7)
    assert nrec == 3
    assert nchars == 8 + 1 + 4 + 1 + 1 + 1        # residues + one terminator per record (SequencePreprocessor.cpp:12)
    assert packed.size == synth.SEGMENT // 4      # padded to one 12288-symbol segment
    sym = synth.unpack_2bit(packed)
    assert sym[:8].tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    assert sym[9:13].tolist() == [2, 2, 3, 3]
    assert sym[14] == 0
    assert not sym[16:].any()                      # padding is symbol 0 = 'A' (SequencePreprocessor.cpp:41)
    # terminators become random nucleotides: rand()%2 is drawn, discarded, then rand()%4 (:74,:82)
    r = rand_stream(libc, 7, 6)
    assert [int(sym[8]), int(sym[13]), int(sym[15])] == [r[1] % 4, r[3] % 4, r[5] % 4]


def test_ambiguity_codes_follow_rand_call_order(tmp_path, libc):
    p = tmp_path / "amb.fa"
    p.write_text(">x\nRSWKMNBDHVrswkmn\n")
    packed, nchars, _ = havac.pack_fasta(str(p), seed=11)
    sym = synth.unpack_2bit(packed)[:nchars]
    r = iter(rand_stream(libc, 11, 64))
    want = []
    for ch in "RSWKMNBDHVRSWKMN\0":
        coin = next(r) % 2
        want.append({"R": coin * 2, "S": coin + 1, "W": coin * 3, "K": coin + 2, "M": coin}.get(ch))
        if want[-1] is None:
            want[-1] = next(r) % 4
    assert sym.tolist() == want


def test_y_is_effectively_a(tmp_path):
    """'Y' returns coin << 2 (0 or 4, SequencePreprocessor.cpp:77): the stray bit lands in the next
    symbol's field and is cleared by that symbol's own write, so Y always packs as A."""
    p = tmp_path / "y.fa"
    p.write_text(">x\n" + "YT" * 50 + "\n")
    for seed in range(4):
        packed, nchars, _ = havac.pack_fasta(str(p), seed=seed)
        sym = synth.unpack_2bit(packed)[:100]
        assert sym[0::2].tolist() == [0] * 50 and sym[1::2].tolist() == [3] * 50


def test_fasta_missing_file():
    with pytest.raises(RuntimeError):
        havac.pack_fasta("/nonexistent/none.fa")


# ---------------------------------------------------------------- HMMER3 reader + projection

def reference_scaling_factor(mu, lam, maxl, L, p):
    """Independent numpy restatement of PhmmReprojection.cpp:36-64 with the same float/double widths."""
    f32, f64 = np.float32, np.float64
    mu, lam, maxl, L, p = f32(mu), f32(lam), f32(maxl), f32(L), f32(p)
    pd = f64(p)
    inner = (pd ** pd - 1) / pd if pd < 5e-9 else np.log(-1.0 * np.log(1.0 - pd))
    inv = f64(mu) - inner / f64(lam)
    n_loop = f32(np.log(f32(maxl / f32(maxl + f32(3)))))
    n_loop_total = f32(n_loop * maxl)
    n_escape = f32(np.log(f32(f32(3) / f32(maxl + f32(3)))))
    b_mk = f32(np.log(f32(f32(2) / f32(L * f32(L + f32(1))))))
    e_c = f32(np.log(f32(0.5)))
    core = f32(f32(f32(f32(n_escape + n_loop_total) + n_escape) + b_mk) + e_c)
    bg_p = f32(maxl / f32(maxl + f32(1)))
    bg_total = f32(maxl * f32(np.log(bg_p)))
    bg_move = f32(np.log(f64(1.0) - f64(bg_p)))
    bg = f32(bg_total + bg_move)
    ln2 = f64(0.69314718055994529)
    thr_nats = f32(inv * ln2 + f64(bg) - f64(core))
    thr_bits = f32(f64(thr_nats) / ln2)
    return f32(f32(256) / thr_bits)


@pytest.mark.parametrize("mu,lam,maxl,L,p", [(-9.0, 0.72, 400, 100, 0.02), (-8.3, 0.70, 1300, 1024, 0.02),
                                             (-10.1, 0.69, 25000, 20000, 0.05), (-7.5, 0.71, 120, 50, 1e-4),
                                             (-9.0, 0.72, 400, 100, 1e-10)])
def test_scaling_factor_widths(mu, lam, maxl, L, p):
    got = havac.scaling_factor(mu, lam, maxl, L, p)
    want = float(reference_scaling_factor(mu, lam, maxl, L, p))
    # exact: numpy's float32 log and glibc's logf agree on these inputs, and the known answers captured from the
    # reference's own code (tests/test_projection_golden.py) hold the product to the bit
    assert np.float32(got).tobytes() == np.float32(want).tobytes()


def test_worked_example_of_the_survey():
    """L=100, MAXL=400, mu=-9, lambda=0.72, p=0.02 -> scale ~14.2, a 2-bit match ~ +28,
    a -ln p = 3 mismatch ~ -33."""
    s = havac.scaling_factor(-9.0, 0.72, 400, 100, 0.02)
    assert 14.0 < s < 14.4
    assert 256.0 / s == pytest.approx(18.1, abs=0.3)        # threshold in bits
    assert havac.project_score(0.0, s) == round(2 * s)       # p = 1 -> 2 bits over the 1/4 background
    assert havac.project_score(3.0, s) in (-33.0, -32.0, -34.0)
    assert havac.project_score(50.0, s) == -128.0 and havac.project_score(-50.0, s) == 127.0   # saturation
    assert havac.project_score(float("inf"), s) == -128.0   # '*' in the file


def test_hmm_reader_and_projection_of_a_collection(tmp_path):
    lengths = [50, 200, 77]
    models, tables = [], []
    for k, L in enumerate(lengths):
        _, cons = synth.dfam_like_model(L, 40 + k)
        em = synth.emissions_from_consensus(cons, 50 + k)
        if k == 1:
            em[3, 2] = np.inf                                # a '*' entry
        mu, lam, maxl = -9.0 + 0.3 * k, 0.70 + 0.01 * k, 4 * L + 10
        models.append(dict(name=f"m{k}", acc=f"RF{k:05d}", emissions=em, maxl=maxl, mu=mu, lam=lam))
        scale = np.float32(havac.scaling_factor(mu, lam, maxl, L, 0.02))
        s = np.float32(np.round(em, 5).astype(np.float32))   # the file carries 5 decimals
        alpha, beta = np.float32(2) * scale, np.float32(1.44269504089) * scale
        y = np.float32(alpha - np.float32(s * beta))
        y = np.where(np.isfinite(y), np.sign(y) * np.floor(np.abs(y) + np.float32(0.5)), -128)   # round half away
        tables.append(np.clip(y, -128, 127).astype(np.int8))
    path = tmp_path / "c.hmm"
    synth.write_hmm(str(path), models)
    got, lens = havac.project_hmm(str(path), 0.02)
    assert lens.tolist() == lengths
    want = np.concatenate(tables)                             # back to back, no separator (PhmmPreprocessor.cpp:9-31)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert got[50 + 3, 2] == -128


def exactly_rounded_float32(text):
    """the binary32 nearest the decimal `text` (ties to even), computed with exact rational arithmetic"""
    from fractions import Fraction
    x = Fraction(text)
    c = np.float32(float(x))
    best, best_err = None, None
    for cand in (np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))):
        err = abs(Fraction(float(cand)) - x)
        even = (int(np.array(cand, np.float32).view(np.uint32)) & 1) == 0
        if best is None or err < best_err or (err == best_err and even):
            best, best_err = cand, err
    return np.float32(best)


def test_hmm_reader_converts_decimals_exactly_and_in_parallel(tmp_path):
    """The reader turns plain decimals of up to seven significant digits into floats with one float division (digits /
    10^k) and sends everything else to strtof; both must give the correctly rounded binary32 -- the int8 projection
    rounds at .5 boundaries and a last-bit difference in an emission can flip a score.  Checked against exact rational
    arithmetic on thousands of values in every spelling a file can hold, in a file big enough (> 256 KB, hundreds of
    models) for the records to be parsed on several threads, and the model order must survive that."""
    rng = np.random.default_rng(123)
    texts = []
    for _ in range(40000):
        kind = rng.integers(0, 8)
        if kind == 0:
            t = f"{rng.uniform(0, 9):.5f}"                                   # what hmmbuild writes
        elif kind == 1:
            t = f"{rng.uniform(0, 0.001):.5f}"                               # tiny: many leading zeros
        elif kind == 2:
            t = f"{rng.uniform(0, 40):.{int(rng.integers(0, 7))}f}"          # other precisions, integers ("17")
        elif kind == 3:
            t = f"{rng.uniform(0, 20):.9f}"                                  # more than seven digits: strtof
        elif kind == 4:
            t = f"{rng.uniform(0, 20):.4e}"                                  # exponent form: strtof
        elif kind == 5:
            t = "0." + "".join(rng.choice(list("0123456789"), size=int(rng.integers(1, 11))))
        elif kind == 6:
            t = str(int(rng.integers(0, 10_000_000)))                        # up to seven digits, no point
        else:
            t = f"{rng.integers(0, 99)}."                                    # trailing point
        texts.append(t)
    L = 50
    nmodels = len(texts) // (4 * L)
    values = np.array(texts[: nmodels * 4 * L]).reshape(nmodels, L, 4)
    lines = []
    for m in range(nmodels):
        lines += ["HMMER3/f [3.1b2 | February 2015]", f"NAME  m{m}", f"LENG  {L}", "MAXL  300", "ALPH  DNA",
                  "STATS LOCAL MSV       -9.1000  0.71000", "HMM          A        C        G        T   ",
                  "            m->m     m->i     m->d     i->m     i->i     d->m     d->d",
                  "  COMPO   1.38629  1.38629  1.38629  1.38629", "          1.38629  1.38629  1.38629  1.38629",
                  "          0.01005  5.29832  5.29832  0.61958  0.77255  0.00000        *"]
        for k in range(L):
            lines += [f"{k + 1:7d}   " + "  ".join(values[m, k]) + f" {k + 1:6d} a - - -",
                      "          1.38629  1.38629  1.38629  1.38629",
                      "          0.01005  5.29832  5.29832  0.61958  0.77255  0.48576  0.95510"]
        lines.append("//")
    path = tmp_path / "decimals.hmm"
    path.write_text("\n".join(lines) + "\n")
    assert path.stat().st_size > (1 << 18)                                   # big enough for the threaded path
    got, count = havac.read_hmm_emissions(str(path))
    assert count == nmodels and got.shape == (nmodels * L, 4)
    want = np.array([exactly_rounded_float32(t) for t in values.reshape(-1)], np.float32).reshape(-1, 4)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))         # bit for bit, model order kept


def test_hmm_format_errors(tmp_path):
    bad = tmp_path / "bad.hmm"
    bad.write_text("this is not a profile\n")
    with pytest.raises(RuntimeError):
        havac.project_hmm(str(bad))
    with pytest.raises(RuntimeError):
        havac.project_hmm(str(tmp_path / "missing.hmm"))
    _, cons = synth.dfam_like_model(10, 1)
    trunc = tmp_path / "trunc.hmm"
    synth.write_hmm(str(trunc), [dict(name="t", acc="", emissions=synth.emissions_from_consensus(cons, 2), maxl=50, mu=-9, lam=0.7)])
    text = trunc.read_text().splitlines()
    trunc.write_text("\n".join(text[:-8]) + "\n")            # cut inside the body, no "//"
    with pytest.raises(RuntimeError):
        havac.project_hmm(str(trunc))
    good = "\n".join(text) + "\n"
    first_node = next(i for i, l in enumerate(text) if l.split()[:1] == ["1"] and len(l.split()) > 5)
    def with_line(i, new_line):
        return "\n".join(text[:i] + [new_line] + text[i + 1:]) + "\n"
    tokens = text[first_node].split()
    cases = {
        "garbage_between": good + "stray line\n" + good,                      # something between two records
        "garbage_before": "stray\n" + good,
        "bad_number": with_line(first_node, " ".join([tokens[0], "1.2x3"] + tokens[2:])),   # a match emission that is not a number
        "too_few_scores": with_line(first_node, " ".join(tokens[:3])),
        "wrong_node_number": with_line(first_node + 3, text[first_node + 3].replace("2", "7", 1)),
        "leng_too_big": good.replace("LENG  10", "LENG  4000000000"),
        "no_leng": good.replace("LENG  10\n", ""),
    }
    for name, broken in cases.items():
        f = tmp_path / (name + ".hmm")
        f.write_text(broken)
        with pytest.raises(RuntimeError):
            havac.project_hmm(str(f))
    ok = tmp_path / "blank_lines.hmm"
    ok.write_text("\n\n" + good + "\n   \n" + good.replace("\n", "\r\n"))            # blank lines around records, CRLF
    assert havac.project_hmm(str(ok))[1].tolist() == [10, 10]


# ---------------------------------------------------------------- resolver

def test_resolver_maps_records_to_hits(tmp_path, oracle):
    fa, hmm = tmp_path / "r.fa", tmp_path / "r.hmm"
    synth.write_fasta(str(fa), [("a", np.zeros(100, np.uint8)), ("b", np.ones(20000, np.uint8)), ("c", np.zeros(7, np.uint8))])
    models = []
    for k, L in enumerate([30, 40]):
        _, cons = synth.dfam_like_model(L, k)
        models.append(dict(name=f"m{k}", acc="", emissions=synth.emissions_from_consensus(cons, k), maxl=200, mu=-9, lam=0.7))
    synth.write_hmm(str(hmm), models)
    # global columns: a = [0,100] (terminator at 100), b = [101, 20101], c = [20102, 20109]; then padding
    rows = [0, 29, 30, 69, 5, 5, 5, 10]
    cols = [0, 100, 101, 20101, 20102, 20108, 20109, 20110]      # the last one is padding -> dropped
    raw = oracle.pack_hits(rows, cols)
    hits = havac.resolve_hits(str(fa), str(hmm), raw)
    got = [(h.sequencePosition, h.sequenceIndex, h.phmmPosition, h.phmmIndex) for h in hits]
    assert got == [(0, 0, 0, 0), (100, 0, 29, 0), (0, 1, 0, 1), (20000, 1, 39, 1), (0, 2, 5, 0), (6, 2, 5, 0), (7, 2, 5, 0)]
    assert hits[1].toString() == "sequence $0, position 100; phmm #0 position 29"


def test_fasta_reader_against_a_plain_python_parse(tmp_path):
    """Long lines (crossing the reader's 1 MiB buffer), CRLF, blank lines, blanks inside lines, no final newline."""
    rng = np.random.default_rng(9)
    alphabet = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    records, text = [], []
    for k in range(7):
        n = int(rng.integers(0, 1_500_000)) if k != 3 else 0
        seq = alphabet[rng.integers(0, alphabet.size, size=n)].tobytes().decode()
        records.append(seq)
        text.append(f">rec{k} some description\r\n" if k % 2 else f">rec{k}\n")
        width = [60, 70, 1_200_000, 80, 61, 999_983, 50][k]
        for i in range(0, n, width):
            line = seq[i:i + width]
            if k == 4 and i == 0 and len(line) > 10:
                line = line[:5] + " \t" + line[5:]          # blanks are squeezed out
            text.append(line + ("\r\n" if k % 2 else "\n"))
        if k == 2:
            text.append("\n\n")
    p = tmp_path / "big.fa"
    p.write_text("".join(text).rstrip("\n"))
    packed, nchars, nrec = havac.pack_fasta(str(p), seed=3)
    assert nrec == 7
    assert nchars == sum(len(r) + 1 for r in records)
    sym = synth.unpack_2bit(packed)
    lut = np.full(256, 255, np.uint8)
    for ch, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[ch] = v
    at = 0
    for r in records:
        want = lut[np.frombuffer(r.encode(), dtype=np.uint8)]
        assert np.array_equal(sym[at:at + len(r)], want)
        at += len(r) + 1                                    # the terminator column holds a random nucleotide


def _windows_by_hand(hits, model_lengths, record_lengths, flank):
    """the definition in Havac.hpp (havacMergeHitsToWindows), written as a union of stretches on a position set"""
    from collections import defaultdict
    covered = defaultdict(set)
    members = defaultdict(list)
    for h in hits:
        n, L = record_lengths[h.sequenceIndex], model_lengths[h.phmmIndex]
        i, k = min(h.sequencePosition, n - 1), h.phmmPosition
        before, after = (L - 1 - k, k) if h.reverseStrand else (k, L - 1 - k)
        lo, hi = max(0, i - before - flank), min(n - 1, i + after + flank)
        key = (h.sequenceIndex, h.reverseStrand, h.phmmIndex)
        covered[key].update(range(lo, hi + 1))
        members[key].append((lo, k))
    out = []
    for key in sorted(covered):
        pos = sorted(covered[key])
        runs, start = [], pos[0]
        for a, b in zip(pos, pos[1:] + [None]):
            if b is None or b != a + 1:
                runs.append((start, a))
                start = b
        for lo, hi in runs:
            inside = [k for (s, k) in members[key] if lo <= s <= hi]
            out.append(havac.HavacWindow(key[0], key[2], key[1], lo, hi, min(inside), max(inside), len(inside)))
    return out


def test_hits_merge_into_windows():
    """SURVEY.md section 8 row f3 (second half): stretches along the hit diagonals, merged per (record, strand, model)."""
    rng = np.random.default_rng(8)
    model_lengths = [30, 120, 7]
    record_lengths = [500, 40, 2000, 1]
    for flank in (0, 5, 60):
        hits = []
        for _ in range(300):
            m = int(rng.integers(0, 3))
            r = int(rng.integers(0, 4))
            hits.append(havac.HavacHit(int(rng.integers(0, record_lengths[r] + 1)), r, int(rng.integers(0, model_lengths[m])),
                                       m, bool(rng.integers(0, 2))))
        got = havac.merge_windows(hits, model_lengths, record_lengths, flank)
        assert got == _windows_by_hand(hits, model_lengths, record_lengths, flank)
        assert sum(w.hitCount for w in got) == len(hits)
        keys = [(w.sequenceIndex, w.reverseStrand, w.phmmIndex, w.sequenceStart) for w in got]
        assert keys == sorted(keys)
        for a, b in zip(got, got[1:]):                          # windows of one key neither overlap nor touch
            if (a.sequenceIndex, a.reverseStrand, a.phmmIndex) == (b.sequenceIndex, b.reverseStrand, b.phmmIndex):
                assert b.sequenceStart > a.sequenceEnd + 1
    # one forward hit: model position 10 of 30 at record position 100 covers [90, 119]; the mirrored stretch on the
    # reverse strand is [81, 110]
    one = havac.merge_windows([havac.HavacHit(100, 0, 10, 0)], model_lengths, record_lengths)
    assert (one[0].sequenceStart, one[0].sequenceEnd, one[0].hitCount) == (90, 119, 1)
    rev = havac.merge_windows([havac.HavacHit(100, 0, 10, 0, True)], model_lengths, record_lengths)
    assert (rev[0].sequenceStart, rev[0].sequenceEnd, rev[0].reverseStrand) == (81, 110, True)
    assert havac.merge_windows([], model_lengths, record_lengths) == []
    # hits naming a model or record that does not exist are dropped, not an error
    assert havac.merge_windows([havac.HavacHit(1, 9, 0, 0), havac.HavacHit(1, 0, 0, 9)], model_lengths, record_lengths) == []


def test_text_and_patches_reproduce_the_host_packer(tmp_path):
    """what loadSequence hands to the GPU packer -- the text plus (column, symbol) for every
    character that is not a/c/g/t -- gives the host packer's bytes when a/c/g/t are packed plainly and the patches
    applied, for the same rand() seed.  (A record's terminator always follows its last residue, so the bit a 'Y'
    spills into the next field is always cleared by that next character, as in the host packer.)"""
    rng = np.random.default_rng(3)
    alphabet = list("ACGTacgtNRYKMSWBDHVnry")
    records = [(f"s{k}", "".join(rng.choice(alphabet, size=int(rng.integers(1, 5000)), p=[.2] * 4 + [.02] * 4 + [.12 / 14] * 14)))
               for k in range(7)]
    records.append(("last", "ACGTY"))                      # 'Y' as the last residue of the file
    fa = tmp_path / "amb.fa"
    synth.write_fasta(str(fa), records)
    code = np.zeros(256, np.uint8)
    for ch, v in zip("ACGT", range(4)):
        code[ord(ch)] = code[ord(ch.lower())] = v
    for seed in range(8):
        packed, nchars, nrec = havac.pack_fasta(str(fa), seed=seed)
        chars, cols, syms = havac.text_and_patches(str(fa), seed=seed)
        assert chars.size == nchars and (np.diff(cols.astype(np.int64)) > 0).all() and syms.max() <= 3
        sym = np.zeros(packed.size * 4, np.uint8)
        sym[:chars.size] = code[chars]
        sym[cols.astype(np.int64)] = syms
        assert np.array_equal(synth.pack_2bit(sym), packed), seed
        plain = code[chars] != 0
        plain |= np.isin(chars, np.frombuffer(b"Aa", np.uint8))
        assert np.array_equal(np.flatnonzero(~plain), cols[cols < nchars].astype(np.int64))   # exactly the other characters
        assert cols[-1] == nchars - 1                         # the last patch is the last record's terminator
