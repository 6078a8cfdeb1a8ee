"""Kernel time of tall-model workloads with and without row blocks (bench.py --tuning).  python tools/split_probe.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rpb in ("0", "4096", "8192", "16384"):
    env = dict(os.environ)
    for w in ("c5", "c3"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--no-pmc", "--no-cpu-baseline", "--steps", "4", "--warmup", "1", f"--tuning={rpb},-1,-1,-1"],
                           capture_output=True, text=True, env=env)
        import json
        try:
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
            print(f"rows per block {rpb:>6} {w}: kernel {d['kernel']['avg_ms']:.2f} ms = {d['kernel']['gcups_kernel_only']/1e3:.1f} TCUPS, step {d['ms_per_step']:.2f} ms", flush=True)
        except Exception as e:
            print(rpb, w, "failed", r.stderr[-300:])
