#!/bin/bash
# A/B of builds of libhavac_dev.so on ONE GPU box, interleaved (boxes differ by ~1 %):
#   build the variants into tools/_bin/ab/libA.so, tools/_bin/ab/libB.so, ... then
#   gpurun -- 'bash tools/ab.sh A B [C ...]'          -> gpurun_out/ab.log
set -e
cd "$(dirname "$0")/.."
variants="${@:-A B}"
cp havac_amd/libhavac_dev.so tools/_bin/ab/_kept.so
: > gpurun_out/ab.log
for round in 1 2; do
  for v in $variants; do
    cp tools/_bin/ab/lib$v.so havac_amd/libhavac_dev.so
    echo "== $v" >> gpurun_out/ab.log
    timeout -k 10 60 python tools/hit_density_probe.py 2>/dev/null | tail -3 >> gpurun_out/ab.log
  done
done
cp tools/_bin/ab/_kept.so havac_amd/libhavac_dev.so
cat gpurun_out/ab.log
