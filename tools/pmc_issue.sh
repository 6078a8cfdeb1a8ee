#!/bin/bash
# Instruction-fetch and LDS-queue counters of ssv_diag_kernel (is the hot loop starved by the instruction cache or by the
# LDS command queue?).  On the GPU box from the repo root:   bash tools/pmc_issue.sh [c2|c3|c5] [dfam|nohit] [tag]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
workload=${1:-c2}; kind=${2:-nohit}; tag=${3:-issue_$kind}
out=gpurun_out/pmc_$tag
run() {
    name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $out -o "$name" -- python3 tools/pmc_probe.py $workload $kind 3 > "${out}_$name.log" 2>&1
}
mkdir -p $out
run pmc_icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE
run pmc_queues SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LDS_ADDR_CONFLICT
python3 tools/pmc_summary.py $out > ${out}_summary.csv
cat ${out}_summary.csv
