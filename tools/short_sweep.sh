#!/bin/bash
# The resident-table kernel (short models) against the standard one on one GPU box (tools/rows_probe.py: kernel ms by HIP events,
# 100 Mbp, without / with hits):   bash tools/short_sweep.sh [rows ...]   -> gpurun_out/short_sweep.txt
# tuning = rows_per_block,tiles_per_item,block_tails,ordering,parts_log2,split_rounds_x4,short_rows,guide,variant
cd "$(dirname "$0")/.."
rows="${@:-32 64 96 128 160 192 256}"
out=gpurun_out/short_sweep.txt
: > $out
for t in "standard:-1,-1,-1,-1,-1,-1,-1,-1,0" "resident:-1,-1,-1,-1,-1,-1,-1,-1,1" "standard_again:-1,-1,-1,-1,-1,-1,-1,-1,0" "resident_again:-1,-1,-1,-1,-1,-1,-1,-1,1"; do
    echo "== ${t%%:*}  (--tuning=${t#*:})" >> $out
    timeout -k 10 150 python tools/rows_probe.py --tuning=${t#*:} $rows 2>/dev/null >> $out || echo "FAILED" >> $out
done
cat $out
