#!/bin/bash
# like tools/ab.sh, with the model-height sweep (tools/rows_probe.py) as the measurement -> gpurun_out/ab_rows.log
set -e
cd "$(dirname "$0")/.."
variants="${@:-A B}"
cp havac_amd/libhavac_dev.so tools/_bin/ab/_kept.so
: > gpurun_out/ab_rows.log
for v in $variants; do
    cp tools/_bin/ab/lib$v.so havac_amd/libhavac_dev.so
    echo "== $v" >> gpurun_out/ab_rows.log
    timeout -k 10 200 python tools/rows_probe.py 32 64 2>/dev/null >> gpurun_out/ab_rows.log
done
cp tools/_bin/ab/_kept.so havac_amd/libhavac_dev.so
cat gpurun_out/ab_rows.log
