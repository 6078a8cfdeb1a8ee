#!/bin/bash
# like tools/ab.sh, with the model-height sweep (tools/rows_probe.py) as the measurement -> gpurun_out/ab_rows.log
#   ROWS="32 64" TUNING=-1,-1,-1,-1,-1,-1,-1,-1,1 bash tools/ab_rows.sh A B   (libraries tools/_bin/ab/lib<X>.so; each variant twice, interleaved)
set -e
cd "$(dirname "$0")/.."
variants="${@:-A B}"
rows="${ROWS:-32 64}"
cp havac_amd/libhavac_dev.so tools/_bin/ab/_kept.so
: > gpurun_out/ab_rows.log
for pass in 1 2; do
for v in $variants; do
    cp tools/_bin/ab/lib$v.so havac_amd/libhavac_dev.so
    echo "== $v" >> gpurun_out/ab_rows.log
    timeout -k 10 200 python tools/rows_probe.py ${TUNING:+--tuning=$TUNING} $rows 2>/dev/null >> gpurun_out/ab_rows.log || echo FAILED >> gpurun_out/ab_rows.log
done
done
cp tools/_bin/ab/_kept.so havac_amd/libhavac_dev.so
cat gpurun_out/ab_rows.log
