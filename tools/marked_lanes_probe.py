"""how many LANES of a wave show a mark when the slow path is entered?  (A vector replay of one register for
all lanes at once would pay only if several adjacent lanes are marked together.)  From the hit list of C2 itself, on the CPU:
a hit (row p, column s) lies on diagonal d = s - p + rows_padded; its wave's tile is d // 2048, its lane (d % 2048) // 32, its
register ((d % 32) // 2), and the four-step window it is found in is (p - half) // 4 (the high cell of a register runs one row
behind).  Entries of the slow path = distinct (tile, window); lanes per entry = distinct lanes among its hits.
python tools/marked_lanes_probe.py [rows]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from havac_amd import synth  # noqa: E402
from oracle import pyoracle as O  # noqa: E402
import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model, packed, ncols, _, planted = bench.make_inputs("c2", 1, rows, 0)
hits = O.ssv_fast(synth.unpack_2bit(packed), model, nthreads=8, cap=1 << 22)
p, s = O.unpack_hits(hits)
p, s = p.astype(np.int64), s.astype(np.int64)
rows_padded = -(-rows // 32) * 32
d = s - p + rows_padded
tile, lane, reg, half = d // 2048, (d % 2048) // 32, (d % 32) // 2, d % 2
window = (p + half) // 4          # the step at which the cell is computed: low cell at step p, high cell at step p + 1
entry = tile * (rows_padded // 4 + 2) + window
order = np.lexsort((lane, entry))
entry, lane, reg = entry[order], lane[order], reg[order]
new_entry = np.r_[True, entry[1:] != entry[:-1]]
new_lane = new_entry | np.r_[True, lane[1:] != lane[:-1]]
entries = int(new_entry.sum())
lanes_per_entry = np.add.reduceat(new_lane.astype(np.int64), np.flatnonzero(new_entry))
print(f"{rows} rows x {ncols} columns: {hits.size} hits, {planted} planted homologs; slow-path entries (tile, window): {entries}; "
      f"hits per entry {hits.size / entries:.3f}; marked lanes per entry: mean {lanes_per_entry.mean():.4f}, "
      f"1 lane {np.mean(lanes_per_entry == 1):.4%}, 2 lanes {np.mean(lanes_per_entry == 2):.4%}, 3 or more {np.mean(lanes_per_entry >= 3):.4%}")
