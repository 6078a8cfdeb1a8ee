"""python tools/collect_profiles.py <tag in gpurun_out> <tag in profiles> "<note>": copies what tools/measure_round.sh left under
gpurun_out/ into profiles/ -- bench lines, the rocprofv3 --kernel-trace --stats summaries (kernel names shortened, the command
and the note in a header line), the PMC summaries, the model-height sweep, the end-to-end series, the PCIe-inclusive rate."""
import csv
import os
import shutil
import sys

src_tag, dst_tag, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def copy(src, dst):
    if os.path.isfile(os.path.join(G, src)) and os.path.getsize(os.path.join(G, src)):
        shutil.copyfile(os.path.join(G, src), os.path.join(P, dst))
        print("profiles/" + dst)


for w in ("c2", "c3", "c5"):
    copy(f"{src_tag}_bench_{w}.json", f"{dst_tag}_bench_{w}.json")
    copy(f"{src_tag}_bench_{w}_under_rocprof.json", f"{dst_tag}_bench_{w}_under_rocprof.json")
    stats = os.path.join(G, f"{src_tag}_prof_{w}", f"{w}_kernel_stats.csv")
    if os.path.isfile(stats):
        rows = list(csv.reader(open(stats)))
        out = os.path.join(P, f"{dst_tag}_kernel_stats_{w}.csv")
        with open(out, "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {w} --no-pmc --no-cpu-baseline (MI355X; {note}); "
                    "every launch of the run is in the average, warm-up and the strictly serial passes included; with passes in flight consecutive SSV kernels "
                    "overlap (two kernel streams) and their durations include the neighbour's share of the chip; kernel names shortened\n")
            trace = os.path.join(G, f"{src_tag}_prof_{w}", f"{w}_kernel_trace.csv")
            if os.path.isfile(trace):      # the same trace, launches that ran alone apart from those that overlapped a neighbour
                ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace))
                            if "ssv_diag_kernel" in r["Kernel_Name"] or "ssv_resident_kernel" in r["Kernel_Name"])
                d = [(b - a) / 1e6 for a, b in ks]
                late = [i > 0 and ks[i][0] < ks[i - 1][1] for i in range(len(ks))]          # started while its predecessor ran
                early = [i + 1 < len(ks) and ks[i + 1][0] < ks[i][1] for i in range(len(ks))]  # its successor started before it ended
                alone = [d[i] for i in range(1, len(ks)) if not late[i] and not early[i]]
                over = [d[i] for i in range(len(ks)) if late[i]]
                if alone:
                    f.write(f"# SSV kernel, from the kernel trace of this run: {len(alone)} launches that shared the chip with no other SSV kernel (the strictly "
                            f"serial passes, warm-up): mean {sum(alone) / len(alone):.4f} ms, min {min(alone):.4f} ms -- the duration `kernel.avg_ms` / `roofline` state; "
                            + (f"{len(over)} launches that started while their predecessor ran (the pipelined passes): mean {sum(over) / len(over):.4f} ms\n" if over else "none overlapped\n"))
            for r in rows:
                r[0] = r[0].split("(")[0]
                f.write(",".join(r) + "\n")
        print("profiles/" + os.path.basename(out))
    stats = os.path.join(G, f"{src_tag}_prof_{w}_serial", f"{w}_kernel_stats.csv")
    if os.path.isfile(stats):
        out = os.path.join(P, f"{dst_tag}_kernel_stats_{w}_one_pass_in_flight.csv")
        with open(out, "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {w} --pipeline-depth 1 --no-pmc --no-cpu-baseline (MI355X; {note}); "
                    "one pass in flight: every SSV kernel runs alone (the duration `roofline` is computed from); warm-up included; kernel names shortened\n")
            for r in csv.reader(open(stats)):
                r[0] = r[0].split("(")[0]
                f.write(",".join(r) + "\n")
        print("profiles/" + os.path.basename(out))
    pmc = os.path.join(G, f"pmc_{src_tag}_{w}_summary.csv")
    if os.path.isfile(pmc):
        with open(os.path.join(P, f"{dst_tag}_pmc_{w}.csv"), "w") as f:
            f.write(f"# bash tools/pmc_passes.sh {w} dfam (one rocprofv3 --pmc pass per group; tools/pmc_probe.py: 3 launches of the workload "
                    f"through the handle API); averages per launch of havac::ssv_diag_kernel; {note}\n")
            f.write(open(pmc).read())
        print(f"profiles/{dst_tag}_pmc_{w}.csv")
copy(f"{src_tag}_bench_c4_one_gpu.json", f"{dst_tag}_bench_c4_one_gpu.json")
copy(f"{src_tag}_rows_sweep.txt", f"{dst_tag}_rows_sweep_kernel_ms.txt")
copy(f"{src_tag}_rows_sweep_standard_kernel.txt", f"{dst_tag}_rows_sweep_kernel_ms_standard_kernel_forced.txt")
copy(f"{src_tag}_bench_rows64.json", f"{dst_tag}_bench_rows64.json")
stats = os.path.join(G, f"{src_tag}_prof_rows64", "rows64_kernel_stats.csv")
if os.path.isfile(stats):
    with open(os.path.join(P, f"{dst_tag}_kernel_stats_rows64.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --rows 64 --no-pmc --no-cpu-baseline (MI355X; {note}); kernel names shortened\n")
        for r in csv.reader(open(stats)):
            r[0] = r[0].split("(")[0]
            f.write(",".join(r) + "\n")
    print(f"profiles/{dst_tag}_kernel_stats_rows64.csv")
copy(f"{src_tag}_e2e_series.txt", f"{dst_tag}_e2e_series_51Mbp.txt")
copy(f"{src_tag}_pcie_inclusive_c2.txt", f"{dst_tag}_pcie_inclusive_c2.txt")
