#!/bin/bash
# One measurement round on the GPU box: bench lines of c2 / c3 / c5, rocprofv3 kernel stats of each, PMC passes of each,
# the model-height sweep.  bash tools/measure_round.sh <tag> [all]   -> gpurun_out/<tag>_*   (`all`: + C4 on one GPU, the
# end-to-end series, the PCIe-inclusive rate; ~2 minutes of GPU time without, ~4 with)
set -e
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in c2 c3 c5; do
    python3 bench.py --workload $w > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err || echo "bench $w failed"
    tail -c 400 gpurun_out/${tag}_bench_$w.json; echo
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_$w -o $w -- python3 bench.py --workload $w --no-pmc --no-cpu-baseline > gpurun_out/${tag}_bench_${w}_under_rocprof.json 2> gpurun_out/${tag}_prof_$w.err || echo "rocprof $w failed"
    # the kernel ALONE (what `roofline` is computed from): the same command with one pass in flight -- with several, consecutive kernels
    # share the chip (two kernel streams) and the trace's durations include the neighbour's share
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_${w}_serial -o $w -- python3 bench.py --workload $w --pipeline-depth 1 --no-pmc --no-cpu-baseline > gpurun_out/${tag}_bench_${w}_serial_under_rocprof.json 2> gpurun_out/${tag}_prof_${w}_serial.err || echo "rocprof $w serial failed"
    bash tools/pmc_passes.sh $w dfam ${tag}_$w > /dev/null 2>&1 || echo "pmc $w failed"
    cat gpurun_out/pmc_${tag}_${w}_summary.csv | head -40
done
python3 tools/rows_probe.py > gpurun_out/${tag}_rows_sweep.txt 2>/dev/null
cat gpurun_out/${tag}_rows_sweep.txt
# short models: the standard kernel forced (the sweep above takes the library's choice: the resident-table kernel up to 256 rows),
# a bench line and the kernel trace of one short model
python3 tools/rows_probe.py --tuning=-1,-1,-1,-1,-1,-1,-1,-1,0 32 64 96 128 160 192 256 > gpurun_out/${tag}_rows_sweep_standard_kernel.txt 2>/dev/null
cat gpurun_out/${tag}_rows_sweep_standard_kernel.txt
python3 bench.py --rows 64 > gpurun_out/${tag}_bench_rows64.json 2> gpurun_out/${tag}_bench_rows64.err || echo "bench rows 64 failed"
tail -c 300 gpurun_out/${tag}_bench_rows64.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_rows64 -o rows64 -- python3 bench.py --rows 64 --no-pmc --no-cpu-baseline > gpurun_out/${tag}_bench_rows64_under_rocprof.json 2> gpurun_out/${tag}_prof_rows64.err || echo "rocprof rows 64 failed"
if [ "$2" = "all" ]; then      # + C4 on one GPU (a 9 s launch per pass), the end-to-end series, the PCIe-inclusive rate
    python3 bench.py --workload c4 --no-pmc > gpurun_out/${tag}_bench_c4_one_gpu.json 2> gpurun_out/${tag}_bench_c4.err || echo "bench c4 failed"
    tail -c 300 gpurun_out/${tag}_bench_c4_one_gpu.json; echo
    python3 tools/e2e_series.py > gpurun_out/${tag}_e2e_series.txt 2>&1 || echo "e2e series failed"
    tail -14 gpurun_out/${tag}_e2e_series.txt
    python3 tools/pcie_inclusive.py 2>/dev/null | tee gpurun_out/${tag}_pcie_inclusive_c2.txt
fi
