#!/bin/bash
# VALU / SALU instructions of ssv_diag_kernel with and without hits (tools/hit_density_probe.py: 25 launches each of
# dfam, nohit, fewhit) -> gpurun_out/pmc_hits.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_hits -o h -- python3 tools/hit_density_probe.py > gpurun_out/pmc_hits.log 2>&1
python3 - <<'PY' > gpurun_out/pmc_hits.txt
import csv, collections
per = collections.defaultdict(dict)
with open("gpurun_out/pmc_hits/h_counter_collection.csv") as f:
    for row in csv.DictReader(f):
        if "ssv_diag_kernel" in row["Kernel_Name"]:
            d = per[int(row["Dispatch_Id"])]
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
ids = sorted(per)
groups = [ids[i * 25:(i + 1) * 25] for i in range(3)]
for name, g in zip(("dfam", "nohit", "fewhit"), groups):
    tail = g[5:]
    avg = {k: sum(per[i][k] for i in tail) / len(tail) for k in per[tail[0]]}
    print(name, {k: round(v) for k, v in sorted(avg.items())})
PY
cat gpurun_out/pmc_hits.txt
