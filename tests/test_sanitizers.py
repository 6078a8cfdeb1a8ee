"""AddressSanitizer + UBSan over the host stages and the CPU checker (CPU builds only: GPU sanitizers are not
available on the MI355X pool).  Malformed and odd inputs included."""
import os
import subprocess

import numpy as np
import pytest

from havac_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "havac_amd", "csrc", "host")
FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]


def build(cmd, out):
    r = subprocess.run(cmd + ["-o", out], capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr or "ubsan" in r.stderr or "sanitize" in r.stderr):
        pytest.skip("this compiler has no sanitizer runtime")
    assert r.returncode == 0, r.stderr
    return out


def test_host_stages_under_asan_ubsan(tmp_path):
    rng = np.random.default_rng(2)
    alphabet = list("ACGTacgtNRYKMSWBDHVnry")
    records = [(f"s{k} desc", "".join(rng.choice(alphabet, size=int(rng.integers(0, 3000))))) for k in range(9)]
    synth.write_fasta(str(tmp_path / "a.fa"), records)
    (tmp_path / "b.fa").write_bytes(b">x\r\nACGT\r\n\r\nAC\r\n>y\n>z\nGGG")       # CRLF, an empty record, no final newline
    (tmp_path / "c.fa").write_bytes(b"")
    (tmp_path / "d.fa").write_bytes(b"ACGT\n>late\nAC\n")                         # residues before any header
    models = []
    for k, length in enumerate([5, 120, 33]):
        _, cons = synth.dfam_like_model(length, 70 + k)
        models.append(dict(name=f"fam{k}", acc=f"RF{k:05d}", emissions=synth.emissions_from_consensus(cons, 80 + k),
                           maxl=3 * length + 50, mu=-9.2, lam=0.71))
    synth.write_hmm(str(tmp_path / "m.hmm"), models)
    text = (tmp_path / "m.hmm").read_text()
    (tmp_path / "cut.hmm").write_text(text[:700])                                   # truncated inside a model
    (tmp_path / "cut2.hmm").write_text(text[: len(text) // 2])
    # a protein model (20 scores per node): the reader accepts it, the preprocessors must refuse it instead of
    # writing 20 * L scores into a 4 * L table
    amino = ["HMMER3/f [3.1b2 | February 2015]", "NAME  prot", "LENG  7", "MAXL  60", "ALPH  amino",
             "STATS LOCAL MSV       -8.1000  0.71000", "HMM          " + "        ".join("ACDEFGHIKLMNPQRSTVWY"),
             "            m->m     m->i     m->d     i->m     i->i     d->m     d->d",
             "  COMPO   " + "  ".join(["2.99573"] * 20), "          " + "  ".join(["2.99573"] * 20),
             "          0.01005  5.29832  5.29832  0.61958  0.77255  0.00000        *"]
    for k in range(7):
        amino += [f"{k + 1:7d}   " + "  ".join(f"{v:7.5f}" for v in rng.uniform(1.0, 5.0, size=20)) + f" {k + 1:6d} a - - -",
                  "          " + "  ".join(["2.99573"] * 20),
                  "          0.01005  5.29832  5.29832  0.61958  0.77255  0.48576  0.95510"]
    (tmp_path / "amino.hmm").write_text("\n".join(amino + ["//"]) + "\n")
    exe = build(["g++", "-std=c++17", "-pthread", "-ffp-contract=off", *FLAGS, "-I" + HOST, os.path.join(ROOT, "tests", "native", "host_sanitize.cpp")] +
                [os.path.join(HOST, f) for f in ("FastaVector.cpp", "p7HmmReader.cpp", "SequencePreprocessor.cpp",
                                                 "PhmmPreprocessor.cpp", "PhmmReprojection.cpp")], str(tmp_path / "host_san"))
    names = ["a.fa", "b.fa", "c.fa", "d.fa", "missing.fa", "m.hmm", "cut.hmm", "cut2.hmm", "missing.hmm", "amino.hmm"]
    r = subprocess.run([exe] + [str(tmp_path / n) for n in names], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
    out = r.stdout
    assert "b.fa: rc 0, 12 chars, 3 records" in out        # 6 + 0 + 3 residues, one terminator each; '\\r' dropped
    assert "c.fa: rc 0, 0 chars, 0 records" in out
    assert "m.hmm: rc 0" in out and "3 models" in out
    assert "amino.hmm: rc 0" in out and "rejected: model 0 of the phmm file is not a nucleotide model" in out
    assert "cut.hmm: rc 3" in out and "missing.hmm: rc 1" in out and "missing.fa: rc 1" in out


def test_oracle_under_asan_ubsan(tmp_path):
    exe = build(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", *FLAGS, "-I" + os.path.join(ROOT, "oracle"),
                 os.path.join(ROOT, "tests", "native", "oracle_sanitize.c"), os.path.join(ROOT, "oracle", "ssv_oracle.c"),
                 "-lpthread"], str(tmp_path / "oracle_san"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "oracle sanitizer run ok" in r.stdout, (r.stdout + r.stderr)[-2000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-2000:]
