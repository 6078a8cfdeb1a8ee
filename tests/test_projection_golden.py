"""The int8 score projection against known answers captured from the reference's PhmmReprojection.cpp
(tests/golden/g7_projection.npz; generator: tests/golden/make_golden_g7.py).  Everything here
is exact: one ulp in the scaling factor can flip a rounded int8 score, and that table is the input of the whole path.
Reference: PhmmReprojection/PhmmReprojection.cpp:15-31 (inverse survival), :36-64 (scaling factor), :88-107 (one
score), :109-145 (the table)."""
import os

import numpy as np
import pytest

from havac_amd import havac, synth

G7 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_projection.npz"))


def cases():
    off_e = off_s = 0
    for i in range(G7["L"].size):
        L = int(G7["L"][i])
        em = G7["emissions"][off_e: off_e + 4 * L].reshape(L, 4)
        table = G7["table"][off_e: off_e + 4 * L].reshape(L, 4)
        n_single = 4 * min(L, 16)
        single = G7["single"][off_s: off_s + n_single]
        off_e += 4 * L
        off_s += n_single
        yield dict(i=i, mu=G7["mu"][i], lam=G7["lam"][i], maxl=int(G7["maxl"][i]), L=L, p=G7["p"][i], kind=str(G7["kind"][i]),
                   emissions=em, table=table, scale=G7["scale"][i], invsurv=float(G7["invsurv"][i]), single=single)


CASES = list(cases())


def test_the_fixture_covers_what_survey_8c_asks_for():
    assert len(CASES) >= 20
    assert any(c["p"] < 5e-9 for c in CASES) and any(c["p"] > 5e-9 for c in CASES)         # both branches of :27-29
    assert any((c["table"] == -128).any() for c in CASES) and any((c["table"] == 127).any() for c in CASES)   # saturation
    assert any(np.isinf(c["emissions"]).any() for c in CASES)                               # '*'


@pytest.mark.parametrize("c", CASES, ids=lambda c: f"{c['i']}-{c['kind']}-L{c['L']}")
def test_projection_known_answers(c):
    assert havac.gumbel_invsurv(float(c["p"]), float(c["mu"]), float(c["lam"])) == c["invsurv"]
    scale = np.float32(havac.scaling_factor(c["mu"], c["lam"], c["maxl"], c["L"], c["p"]))
    assert scale.tobytes() == c["scale"].tobytes()                                          # bit for bit
    got = havac.project_model(c["mu"], c["lam"], c["maxl"], c["emissions"], c["p"])
    assert np.array_equal(got, c["table"])
    single = np.array([havac.project_score(v, scale) for v in c["emissions"][:16].ravel()], np.float32)
    assert np.array_equal(single, c["single"])


def test_projection_known_answers_through_the_hmm_file(tmp_path):
    """The same answers through the file-level entry point: the p = 0.02 cases written as one HMMER3/f file (mu with
    four decimals, lambda and the emissions with five, '*' for zero probability: the text the fixture's float32 values
    were taken through), read by the product's reader, concatenated by PhmmPreprocessor."""
    chosen = [c for c in CASES if c["p"] == np.float32(0.02) and c["maxl"] > 1]
    assert len(chosen) >= 10
    synth.write_hmm(str(tmp_path / "g7.hmm"), [dict(name=f"g7_{c['i']}", acc=f"G7{c['i']:05d}", emissions=c["emissions"].astype(np.float64),
                                                    maxl=c["maxl"], mu=float(c["mu"]), lam=float(c["lam"])) for c in chosen])
    table, lengths = havac.project_hmm(str(tmp_path / "g7.hmm"), 0.02)
    assert lengths.tolist() == [c["L"] for c in chosen]
    assert np.array_equal(table, np.concatenate([c["table"] for c in chosen]))


def test_protein_model_is_refused(tmp_path):
    """ALPH amino: 20 scores per node would overrun the 4-per-node table."""
    lines = ["HMMER3/f [3.1b2 | February 2015]", "NAME  prot", "LENG  3", "MAXL  40", "ALPH  amino",
             "STATS LOCAL MSV       -8.1000  0.71000", "HMM          " + "        ".join("ACDEFGHIKLMNPQRSTVWY"),
             "            m->m     m->i     m->d     i->m     i->i     d->m     d->d",
             "  COMPO   " + "  ".join(["2.99573"] * 20), "          " + "  ".join(["2.99573"] * 20),
             "          0.01005  5.29832  5.29832  0.61958  0.77255  0.00000        *"]
    for k in range(3):
        lines += [f"{k + 1:7d}   " + "  ".join(["2.99573"] * 20) + f" {k + 1:6d} a - - -", "          " + "  ".join(["2.99573"] * 20),
                  "          0.01005  5.29832  5.29832  0.61958  0.77255  0.48576  0.95510"]
    (tmp_path / "prot.hmm").write_text("\n".join(lines + ["//"]) + "\n")
    with pytest.raises(ValueError):
        havac.project_hmm(str(tmp_path / "prot.hmm"), 0.02)
