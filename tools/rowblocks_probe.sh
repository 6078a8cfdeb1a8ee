#!/bin/bash
# C2 kernel time with tiles cut into 1..4 row blocks (HAVAC_ROW_BLOCKS forces the count)  -> gpurun_out/rowblocks.log
cd "$(dirname "$0")/.."
: > gpurun_out/rowblocks.log
for k in 1 2 3 4 1 3; do
    echo "== row blocks $k" >> gpurun_out/rowblocks.log
    HAVAC_ROW_BLOCKS=$k timeout -k 10 200 python tools/hit_density_probe.py 2>/dev/null | tail -3 >> gpurun_out/rowblocks.log
done
cat gpurun_out/rowblocks.log
