#!/bin/bash
# Hardware counters of ssv_diag_kernel, one rocprofv3 --pmc pass per counter group (never combined with a trace
# domain).  Run on the GPU box from the repo root:   bash tools/pmc_passes.sh [c2|c3|c5] [dfam|nohit] [tag]
#   -> gpurun_out/pmc_<tag>/*.csv and gpurun_out/pmc_<tag>_summary.csv
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
workload=${1:-c2}; kind=${2:-dfam}; tag=${3:-$workload}
out=gpurun_out/pmc_$tag
run() {   # name, counters...
    name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $out -o "$name" -- python3 tools/pmc_probe.py $workload $kind 3 > "${out}_$name.log" 2>&1
}
mkdir -p $out
run pmc_sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run pmc_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
run pmc_sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES
run pmc_fetch FETCH_SIZE
run pmc_write WRITE_SIZE
python3 tools/pmc_summary.py $out > ${out}_summary.csv
cat ${out}_summary.csv
