#!/bin/bash
# A/B of builds of libhavac_dev.so (tools/_bin/ab/lib<X>.so) on what a caller of the reference's API pays for its hit list: havac_benchmark
# --raw on C2 WITH every run's list copied to the host, one run at a time and two in flight.   bash tools/ab_readback.sh A B   -> gpurun_out/ab_readback.log
set -e
cd "$(dirname "$0")/.."
variants="${@:-A B}"
python tools/dump_workload.py --workload c2 /tmp/c2 > /dev/null
cp havac_amd/libhavac_dev.so tools/_bin/ab/_kept.so
: > gpurun_out/ab_readback.log
for pass in 1 2; do
for v in $variants; do
    cp tools/_bin/ab/lib$v.so havac_amd/libhavac_dev.so
    echo "== $v" >> gpurun_out/ab_readback.log
    for depth in 1 2; do
        timeout -k 10 120 havac_amd/havac_benchmark --raw /tmp/c2.seq /tmp/c2.model --repeat 200 --depth $depth 2>&1 | tail -1 >> gpurun_out/ab_readback.log || echo FAILED >> gpurun_out/ab_readback.log
    done
    timeout -k 10 120 havac_amd/havac_benchmark --raw /tmp/c2.seq /tmp/c2.model --repeat 200 --depth 2 --no-readback 2>&1 | tail -1 >> gpurun_out/ab_readback.log
done
done
cp tools/_bin/ab/_kept.so havac_amd/libhavac_dev.so
cat gpurun_out/ab_readback.log
