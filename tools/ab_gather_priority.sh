#!/bin/bash
# A/B of builds of libhavac_dev.so on a rehearsal of an N-rank run on ONE GPU (bench.py --gpus N --backend gloo over the stand-in for
# librccl): ms per step and the device time of rank 0's gathers.   bash tools/ab_gather_priority.sh A G   -> gpurun_out/ab_gather.log
set -e
cd "$(dirname "$0")/.."
variants="${@:-A G}"
python -c "import sys; sys.path.insert(0, 'tests'); from test_gpu_gather_ranks import build_standin; print(build_standin())" > /tmp/standin_path.txt
standin=$(tail -1 /tmp/standin_path.txt)
cp havac_amd/libhavac_dev.so tools/_bin/ab/_kept.so
: > gpurun_out/ab_gather.log
for pass in 1 2; do
for v in $variants; do
    cp tools/_bin/ab/lib$v.so havac_amd/libhavac_dev.so
    for n in 2 4; do
        echo "== $v, $n ranks" >> gpurun_out/ab_gather.log
        timeout -k 10 300 python bench.py --gpus $n --steps 30 --warmup 3 --rows 1024 --columns-per-gpu 25006080 --backend gloo --gather-library $standin --no-pmc --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'gather_ms_rank0', d['distributed'].get('gather_ms_rank0'), 'per rank gather_ms', [p.get('gather_ms') for p in d['distributed']['per_rank']])
" >> gpurun_out/ab_gather.log || echo FAILED >> gpurun_out/ab_gather.log
    done
done
done
cp tools/_bin/ab/_kept.so havac_amd/libhavac_dev.so
cat gpurun_out/ab_gather.log
