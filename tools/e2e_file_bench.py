"""End-to-end timing of the file-level API (counterpart of the reference's benchmark/benchmark.cpp figure):
writes a synthetic FASTA (default 100 Mbp in 20 records) and a .hmm collection, then runs havac_benchmark on them.
    python tools/e2e_file_bench.py [nsymbols] [total_model_rows]
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from havac_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
d = tempfile.mkdtemp(prefix="havac_e2e_")
fa, hmm = os.path.join(d, "db.fa"), os.path.join(d, "models.hmm")
t0 = time.time()
rng = np.random.default_rng(1)
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
nrec = 20
with open(fa, "wb") as f:
    for k in range(nrec):
        m = n // nrec
        s = letters[rng.integers(0, 4, size=m, dtype=np.uint8)]
        width = 60
        pad = (-m) % width
        body = np.concatenate([s, np.full(pad, ord("A"), np.uint8)]).reshape(-1, width)
        lines = np.concatenate([body, np.full((body.shape[0], 1), ord("\n"), np.uint8)], axis=1)
        f.write(f">chr{k} synthetic\n".encode())
        f.write(lines.tobytes())
models, total, k = [], 0, 0
lengths = synth.model_lengths(2000)
while total < rows:
    L = int(min(lengths[k], rows - total)) or 1
    _, cons = synth.dfam_like_model(L, 500 + k)
    models.append(dict(name=f"fam{k}", acc=f"RF{k:05d}", emissions=synth.emissions_from_consensus(cons, 600 + k),
                       maxl=3 * L + 50, mu=-9.0, lam=0.71))
    total += L
    k += 1
synth.write_hmm(hmm, models)
print(f"wrote {os.path.getsize(fa)/1e6:.0f} MB FASTA ({nrec} records) and {len(models)} models / {total} rows in {time.time()-t0:.1f} s", flush=True)
exe = os.path.join(ROOT, "havac_amd", "havac_benchmark")
for rep in range(2):
    t0 = time.time()
    out = subprocess.run([exe, fa, hmm], capture_output=True, text=True)
    print(f"--- run {rep}: wall {time.time()-t0:.2f} s, rc {out.returncode}")
    print(out.stdout.strip())
    if out.returncode:
        print(out.stderr[-2000:])
