#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of a 1 GiB stream per access width (tools/fetch_calibration.hip), one rocprofv3 --pmc pass per counter.
# On the GPU box, from the repo root:   bash tools/fetch_calibration.sh   -> table on stdout
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p tools/_bin gpurun_out/fetch_cal
[ -x tools/_bin/fetch_calibration ] || hipcc --offload-arch=gfx950 -O3 -o tools/_bin/fetch_calibration tools/fetch_calibration.hip
for counter in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $counter --output-format csv -d gpurun_out/fetch_cal -o $counter -- tools/_bin/fetch_calibration > gpurun_out/fetch_cal/$counter.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/fetch_cal/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        per[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]] += float(row["Counter_Value"])
print("kernel (1 GiB streamed once)      FETCH_SIZE [KB]   x1024 / bytes    WRITE_SIZE [KB]   x1024 / bytes")
for k in sorted(per):
    if "stream_" not in k: continue
    f, w = per[k].get("FETCH_SIZE", 0.0), per[k].get("WRITE_SIZE", 0.0)
    print(f"{k:30s} {f:16.1f} {f * 1024 / 2**30:14.3f} {w:18.1f} {w * 1024 / 2**30:14.3f}")
PY
