import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from havac_amd import synth, _lib
from havac_amd.hw_client import HavacHwClient
import ctypes as C
nseg, nrows = 100, 10_752
n = nseg * synth.SEGMENT
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 4
c = HavacHwClient(deviceIndices=[0] * parts)
per_part = n // parts * (nrows // 3)
c.setHitCapacity(per_part + (1 << 25))
c.writeSequence(np.zeros(n // 4, np.uint8))
c.writePhmm(np.full((nrows, 4), 127, np.int8))
t0 = time.time()
c.invokeHavacSsvAsync()
st = c._L.havac_dev_wait(c._h, 0)
print("state", st, "after", round(time.time() - t0, 2), "s;", (c._L.havac_dev_last_error(c._h) or b"").decode(), flush=True)
print("kernel ms", c.lastRunMs() if st == 4 else None)
