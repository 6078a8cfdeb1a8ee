"""How a launch hands out its tiles (havac_dev_set_tuning) against kernel time and HBM traffic.
python3 tools/split_sweep.py <workload> <tuning> [<tuning> ...]     tuning = eight comma-separated values, or `default`
For every tuning: a plain run of tools/pmc_probe.py (kernel ms, HIP events) and two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE: they do not fit one pass on gfx950); traffic = (2 FETCH_SIZE + WRITE_SIZE) KB per launch (the guide's gfx950
correction).  The parent process never touches the GPU."""
import csv
import glob
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
probe = os.path.join(ROOT, "tools", "pmc_probe.py")
workload = sys.argv[1]
kind = "dfam"
launches = {"c2": "12", "c3": "3", "c5": "4"}.get(workload, "4")


def counter(name, tuning, tag):
    out = os.path.join(ROOT, "gpurun_out", f"sweep_{tag}_{name}")
    shutil.rmtree(out, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--pmc", name, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, probe, workload, kind, "3", tuning],
                       capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    per = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if ("ssv_diag_kernel" in row.get("Kernel_Name", "") or "ssv_resident_kernel" in row.get("Kernel_Name", "")) and row.get("Counter_Name") == name:
                per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    shutil.rmtree(out, ignore_errors=True)
    return sum(per.values()) / len(per) if per else float("nan")


for k, tuning in enumerate(sys.argv[2:]):
    t = "" if tuning == "default" else tuning
    r = subprocess.run([sys.executable, probe, workload, kind, launches, t], capture_output=True, text=True)
    m = re.search(r"hits (\d+) .*after the first ([0-9.]+) min ([0-9.]+)", r.stdout)
    if not m:
        print(tuning, "failed:", r.stdout[-200:], r.stderr[-400:], flush=True)
        continue
    fetch = counter("FETCH_SIZE", t, f"{workload}_{k}")
    write = counter("WRITE_SIZE", t, f"{workload}_{k}")
    print(f"{workload} tuning {tuning:>28}: kernel mean {float(m.group(2)):9.4f} ms, min {float(m.group(3)):9.4f} ms, hits {m.group(1)}, "
          f"FETCH_SIZE {fetch / 1024:9.1f} MB, WRITE_SIZE {write / 1024:9.1f} MB, traffic (2 F + W) {(2 * fetch + write) / 1024:9.1f} MB", flush=True)
