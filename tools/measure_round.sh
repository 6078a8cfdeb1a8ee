#!/bin/bash
# One measurement round on the GPU box: bench lines of c2 / c3 / c5, rocprofv3 kernel stats of each, PMC passes of each,
# the model-height sweep.  bash tools/measure_round.sh <tag>   -> gpurun_out/<tag>_*
set -e
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in c2 c3 c5; do
    python3 bench.py --workload $w > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err || echo "bench $w failed"
    tail -c 400 gpurun_out/${tag}_bench_$w.json; echo
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_$w -o $w -- python3 bench.py --workload $w --no-pmc --no-cpu-baseline > gpurun_out/${tag}_bench_${w}_under_rocprof.json 2> gpurun_out/${tag}_prof_$w.err || echo "rocprof $w failed"
    bash tools/pmc_passes.sh $w dfam ${tag}_$w > /dev/null 2>&1 || echo "pmc $w failed"
    cat gpurun_out/pmc_${tag}_${w}_summary.csv | head -40
done
python3 tools/rows_probe.py > gpurun_out/${tag}_rows_sweep.txt 2>/dev/null
cat gpurun_out/${tag}_rows_sweep.txt
