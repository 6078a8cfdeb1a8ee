#!/bin/bash
# A/B of two builds of libhavac_dev.so on ONE GPU box, interleaved (boxes differ by ~1 %):
#   build the two variants into build/ab/libA.so and build/ab/libB.so, then
#   gpurun -- 'bash tools/ab.sh'          -> gpurun_out/ab.log
set -e
cd "$(dirname "$0")/.."
: > gpurun_out/ab.log
for v in A B A B; do
    cp build/ab/lib$v.so havac_amd/libhavac_dev.so
    echo "== $v" >> gpurun_out/ab.log
    timeout -k 10 200 python tools/hit_density_probe.py 2>/dev/null | tail -3 >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
