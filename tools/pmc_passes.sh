#!/bin/bash
# Hardware counters of ssv_diag_kernel on C2, one rocprofv3 --pmc pass per counter group (never combined with a
# trace domain).  Run on the GPU box from the repo root:   bash tools/pmc_passes.sh   -> gpurun_out/pmc/*.csv
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {   # name, counters...
    name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc -o "$name" -- python3 bench.py --steps 3 --warmup 1 --pipeline-depth 1 --no-cpu-baseline > "gpurun_out/pmc_$name.log" 2>&1
}
run pmc_sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run pmc_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
run pmc_fetch FETCH_SIZE
run pmc_write WRITE_SIZE
python3 tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc_summary.csv
cat gpurun_out/pmc_summary.csv
