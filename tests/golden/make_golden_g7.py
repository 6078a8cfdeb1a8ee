"""Generates tests/golden/g7_projection.npz: known answers of the int8 score projection.

Run in the build container only (needs /root/reference):

    make -C oracle _ref && python tests/golden/make_golden_g7.py

The answers come from the REFERENCE's PhmmReprojection/PhmmReprojection.cpp compiled from where it lies, against the
product's own p7HmmReader.h (the reference's reader is an un-vendored submodule): a differential check with a
substituted header, see oracle/reproj_wrap.cpp.  Two objects are used, -O0 (what the reference's CMakeLists.txt
builds: it sets no C++ optimisation level) and -O2; the script refuses to write anything unless they agree bit for bit.

Each case is data only: (mu, lambda, MAXL, L, p) + the match emissions [L,4] as float32 file values (-ln p, +inf for
'*') -> the float32 scaling factor and the int8 [L,4] table.  A text rendering of the first cases as HMMER3/f is
checked by tests/test_projection_golden.py through the product's own reader and file-level entry point.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))


def load(name):
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", name))
    L.reproj_ref_scale.restype = C.c_float
    L.reproj_ref_scale.argtypes = [C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_float]
    L.reproj_ref_score.restype = C.c_float
    L.reproj_ref_score.argtypes = [C.c_float, C.c_float]
    L.reproj_ref_invsurv.restype = C.c_double
    L.reproj_ref_invsurv.argtypes = [C.c_double, C.c_double, C.c_double]
    L.reproj_ref_project.restype = None
    L.reproj_ref_project.argtypes = [C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p]
    return L


def through_text(values, fmt):
    """the float32 a reader gets from the decimal text the .hmm file holds (strtof = correctly rounded)"""
    return np.array([np.float32(fmt % v) for v in np.asarray(values, dtype=np.float64).ravel()], dtype=np.float32).reshape(np.shape(values))


def main():
    refs = [load("libreprojection_ref_O0.so"), load("libreprojection_ref.so")]
    rng = np.random.default_rng(7007)
    cases = []
    # (mu, lambda, MAXL, L, p, kind of emissions)
    spec = [
        (-9.0, 0.72, 400, 100, 0.02, "dfam"),            # the worked example of SURVEY.md A.5
        (-8.1234, 0.70412, 190, 50, 0.02, "dfam"),
        (-9.8765, 0.71003, 2345, 2000, 0.02, "dfam"),
        (-8.5, 0.69, 777, 300, 0.001, "dfam"),
        (-10.0, 0.73, 1200, 512, 1e-5, "dfam"),
        (-9.25, 0.7, 640, 257, 1e-7, "dfam"),
        (-9.0, 0.72, 400, 100, 4.9e-9, "dfam"),          # just under HAVAC_GUMBEL_EPSILON: the (p^p - 1)/p branch
        (-9.0, 0.72, 400, 100, 5e-9, "dfam"),            # the float nearest 5e-9 against the double constant
        (-9.0, 0.72, 400, 100, 5.1e-9, "dfam"),
        (-8.75, 0.705, 900, 333, 1e-9, "dfam"),
        (-9.5, 0.715, 1500, 999, 1e-12, "dfam"),
        (-9.1, 0.71, 350, 64, 1e-30, "dfam"),
        (-9.0, 0.72, 400, 100, 0.02, "star"),            # '*' = zero probability = +inf file value
        (-9.0, 0.72, 400, 100, 0.02, "extreme"),         # file values from 0 to 40: both saturations
        (-7.0, 0.9, 60, 8, 0.5, "extreme"),              # a small threshold: a large multiplier, +127 reached
        (-9.3, 0.70, 128, 32, 0.02, "uniform"),          # every emission 1.38629 (= 1/4): scores of 0
        (-9.3, 0.70, 128, 32, 0.02, "halves"),           # values laid around .5 boundaries of the projected score
        (-8.9, 0.695, 5000, 1777, 0.02, "dfam"),
        (-9.6, 0.725, 300, 150, 0.1, "dfam"),
        (-9.05, 0.7105, 1000, 1, 0.02, "dfam"),          # L = 1: logf(2 / (1 * 2)) = 0
        (-9.05, 0.7105, 1, 1, 0.02, "dfam"),             # MAXL = 1
        (-9.05, 0.7105, 100000, 2000, 0.02, "dfam"),
        (-12.0, 0.55, 800, 400, 0.02, "dfam"),
        (-6.0, 1.1, 800, 400, 0.02, "dfam"),
    ]
    for mu, lam, maxl, L, p, kind in spec:
        mu32, lam32, p32 = np.float32("%.4f" % mu), np.float32("%.5f" % lam), np.float32(p)
        if kind == "dfam":
            cons = rng.integers(0, 4, size=L)
            pc = rng.uniform(0.55, 0.99, size=L)
            rest = rng.dirichlet(np.ones(3) * 3, size=L) * (1 - pc)[:, None]
            prob = np.empty((L, 4))
            for k in range(L):
                others = [a for a in range(4) if a != cons[k]]
                prob[k, cons[k]] = pc[k]
                prob[k, others] = rest[k]
            em = -np.log(prob)
        elif kind == "star":
            em = rng.uniform(0.05, 6.0, size=(L, 4))
            em[rng.random((L, 4)) < 0.2] = np.inf
        elif kind == "extreme":
            em = rng.uniform(0.0, 40.0, size=(L, 4))
            em[0] = [0.0, 40.0, 1e-5, 39.99999]
        elif kind == "uniform":
            em = np.full((L, 4), 1.38629)
        elif kind == "halves":
            # projected = 2m - s * log2(e) * m; pick s so that the projection lands near k + 0.5
            m = refs[0].reproj_ref_scale(mu32, lam32, maxl, L, p32)
            k = rng.integers(-120, 120, size=(L, 4)) + 0.5
            em = (2 * m - k) / (1.44269504089 * m) + rng.integers(-3, 4, size=(L, 4)) * 1e-5
            em = np.maximum(em, 0.0)
        em32 = through_text(em, "%.5f")
        em32[~np.isfinite(em)] = np.inf
        answers = []
        for R in refs:
            scale = np.float32(R.reproj_ref_scale(mu32, lam32, maxl, L, p32))
            table = np.empty((L, 4), dtype=np.int8)
            R.reproj_ref_project(mu32, lam32, maxl, L, p32, em32.ctypes.data, table.ctypes.data)
            invsurv = R.reproj_ref_invsurv(float(p32), float(mu32), float(lam32))
            single = np.array([R.reproj_ref_score(v, scale) for v in em32[: min(L, 16)].ravel()], dtype=np.float32)
            answers.append((scale, table, invsurv, single))
        a, b = answers
        assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3]), \
            f"-O0 and -O2 builds of the reference disagree on {(mu, lam, maxl, L, p, kind)}"
        # the legacy per-score function (PhmmReprojection.cpp:88-107) and the table loop (:133-143) are different
        # float expressions; record both
        cases.append(dict(mu=mu32, lam=lam32, maxl=maxl, L=L, p=p32, kind=kind, emissions=em32, scale=a[0], table=a[1],
                          invsurv=a[2], single=a[3]))
        print(f"{kind:8s} L={L:5d} MAXL={maxl:6d} mu={mu32:9.4f} lambda={lam32:.5f} p={p32:.3g}: scale {a[0]:.7g}, "
              f"table min {a[1].min()} max {a[1].max()}, {(a[1] == -128).sum()} at -128, {(a[1] == 127).sum()} at 127")
    np.savez_compressed(
        os.path.join(OUT, "g7_projection.npz"),
        mu=np.array([c["mu"] for c in cases], np.float32), lam=np.array([c["lam"] for c in cases], np.float32),
        maxl=np.array([c["maxl"] for c in cases], np.uint32), L=np.array([c["L"] for c in cases], np.uint32),
        p=np.array([c["p"] for c in cases], np.float32), kind=np.array([c["kind"] for c in cases]),
        scale=np.array([c["scale"] for c in cases], np.float32), invsurv=np.array([c["invsurv"] for c in cases], np.float64),
        emissions=np.concatenate([c["emissions"].reshape(-1) for c in cases]),
        table=np.concatenate([c["table"].reshape(-1) for c in cases]),
        single=np.concatenate([c["single"] for c in cases]),
        note=np.array("int8 projection known answers from the reference's PhmmReprojection.cpp (compiled against the "
                      "product's p7HmmReader.h; -O0 and -O2 objects agree); see make_golden_g7.py"))
    print(f"{len(cases)} cases -> g7_projection.npz")


if __name__ == "__main__":
    main()
