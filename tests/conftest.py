import glob
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- first, on purpose: torch ships its own libamdhip64; whichever HIP runtime is
              # loaded first owns the GPU for the process, and libhavac_dev.so binds to the one already loaded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    """the SSV fixtures (packed sequence, model, hits); g7_* holds the projection's known answers (test_projection_golden.py),
    g8_* the reference's per-(model, record) SSV on small files (test_boundary_golden.py)"""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith(("g7_", "g8_"))]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return z["packed"], z["model"], z["hits"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


def g8_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "g8_boundary_*.npz")))


def load_g8(name, tmp_path):
    """-> (fasta path, hmm path, p-value, HitsFromSsv rows as sorted tuples (record, model, position, row), npz)"""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    fa, hmm = tmp_path / (name + ".fa"), tmp_path / (name + ".hmm")
    fa.write_text(str(z["fasta"]))
    hmm.write_text(str(z["hmm"]))
    assert bool(z["single_equals_table"])
    return str(fa), str(hmm), float(z["p"]), sorted(tuple(int(v) for v in row) for row in z["hits"]), z
