"""python tools/jl.py <file with a bench.py JSON line> [label]: the line's headline figures on one line"""
import json
import sys

for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        k = d["kernel"]
        print(" ".join(sys.argv[2:]), d["config"]["workload"][:3], "value", d["value"], "ms/step", d["ms_per_step"], "serial",
              d["config"]["ms_per_step_strictly_serial"], "kernel", k["avg_ms"], "enq->ordered", k["enqueue_to_ordered_ms"],
              "frac", d["roofline"]["frac"], "clock passes", d.get("clock_warmup_passes"))
