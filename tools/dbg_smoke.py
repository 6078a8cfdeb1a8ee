import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from havac_amd import synth
from havac_amd.hw_client import run_ssv
model, cons = synth.dfam_like_model(300, 2024)
sym = synth.random_symbols(3 * synth.SEGMENT - 100, 1024)
synth.plant_homologs(sym, cons, sym.size, every=9000, length=200)
got = run_ssv(synth.pack_2bit(sym), model, device=0)
print("hits", got.size)
